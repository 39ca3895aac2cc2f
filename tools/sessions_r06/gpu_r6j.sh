# round 6, session j: GroupNorm-backward sums in the epilogue of the halo conv's data-gradient launch (tilings 117 / 124 / 125; with the conv_stream form of session i under
# the same SEG_RQ_FUSE switch): tests, the bench with SEG_RQ_FUSE=1 / 0 in alternation, other configs, a kernel trace
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6j; mkdir -p $O
python -c "from pytorchdeeplearing_amd import _capi; print(_capi.product_library().build_info())" > $O/build.txt 2>&1; cat $O/build.txt
timeout 900 python -m pytest tests/test_conv3x.py tests/test_engine.py -m gpu -q -x -k "backward_sums or sums_on_the_data or conv3x_exact or parity_lowp_gpu or parity_f32_gpu" > $O/tests.log 2>&1; tail -3 $O/tests.log
AB="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --roofline-steps 0"
G='"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*'
for i in 1 2 3; do
  for c in 1 0; do
    echo "== SEG_RQ_FUSE=$c ($i)" >> $O/rq_ab.log; SEG_RQ_FUSE=$c timeout 200 $AB 2>/dev/null | grep -o "$G" >> $O/rq_ab.log
  done
done
cat $O/rq_ab.log
SEG_BENCH_ONLY=C4,C5,C2 timeout 300 python tools/bench_configs.py > $O/configs_rq1.jsonl 2> $O/configs_rq1.err
SEG_RQ_FUSE=0 SEG_BENCH_ONLY=C4,C5,C2 timeout 300 python tools/bench_configs.py > $O/configs_rq0.jsonl 2> $O/configs_rq0.err
cut -c1-120 $O/configs_rq1.jsonl $O/configs_rq0.jsonl
rm -rf gpurun_out/trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --roofline-steps 0 --condition-seconds 0.2 --launch stream > $O/trace_run.log 2>&1
CSV=$(find gpurun_out/trace -name "*kernel_trace.csv" | head -1)
if [ -n "$CSV" ]; then python tools/trace_gaps.py $CSV > $O/trace_timeline.txt 2>&1; fi
rm -rf gpurun_out/trace
head -6 $O/trace_timeline.txt; grep -n "conv3x16_kernel\|gn_bwd_reduce" $O/trace_timeline.txt | head -30
