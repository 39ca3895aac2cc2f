# round 6, session g: the activation between the VNet up-conv and the 1^d conv on the concat applied by its readers on load (SEG_VACT; conv_stream_kernel /
# wgrad_direct_kernel <..., ACT>): operator + engine tests, then the bench with SEG_VACT=1 / 0 in alternation (one binary, one call), other configs, kernel trace
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6g; mkdir -p $O
python -c "from pytorchdeeplearing_amd import _capi; print(_capi.product_library().build_info())" > $O/build.txt 2>&1; cat $O/build.txt
timeout 900 python -m pytest tests/test_ops.py tests/test_engine.py -m gpu -q -x -k "activation or abi or wgrad_exact or conv_gather or one_launch or parity_lowp_gpu" > $O/tests.log 2>&1; tail -5 $O/tests.log
AB="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --roofline-steps 0"
G='"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*'
for i in 1 2 3; do
  for c in 1 0; do
    echo "== SEG_VACT=$c ($i)" >> $O/vact_ab.log; SEG_VACT=$c timeout 200 $AB 2>/dev/null | grep -o "$G" >> $O/vact_ab.log
  done
done
cat $O/vact_ab.log
for c in 1 0; do SEG_VACT=$c SEG_BENCH_ONLY=C5,C2 timeout 300 python tools/bench_configs.py > $O/configs_vact$c.jsonl 2> $O/configs_vact$c.err; done
cut -c1-200 $O/configs_vact1.jsonl $O/configs_vact0.jsonl
rm -rf gpurun_out/trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --roofline-steps 0 --condition-seconds 0.2 --launch stream > $O/trace_run.log 2>&1
CSV=$(find gpurun_out/trace -name "*kernel_trace.csv" | head -1)
if [ -n "$CSV" ]; then python tools/trace_gaps.py $CSV > $O/trace_timeline.txt 2>&1; fi
rm -rf gpurun_out/trace
head -6 $O/trace_timeline.txt; grep -n "conv_stream_kernel\|wgrad_direct" $O/trace_timeline.txt | head -20
