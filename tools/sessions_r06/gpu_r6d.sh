# round 6, session d: (1) the one-launch GroupNorm backward of the >= 64-channel levels (gn_bwd_coop_kernel): equivalence test on the GPU, then the bench
# with SEG_GN_COOP=1 / 0 in alternation (one binary, one call); (2) which weight-gradient launches share a reduce (diagnostic variant, SEG_DIAG_W3_MODE
# 0 = none / 1 = a level visit / 2 = the layers released together); (3) kernel trace of the new binary
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6d; mkdir -p $O
python -c "from pytorchdeeplearing_amd import _capi; print(_capi.product_library().build_info())" > $O/build.txt 2>&1; cat $O/build.txt
timeout 900 python -m pytest tests/test_engine.py -m gpu -q -x -k "one_launch_groupnorm or parity_lowp_gpu or parity_f32_gpu" > $O/coop_tests.log 2>&1; tail -5 $O/coop_tests.log
AB="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --roofline-steps 0"
G='"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*'
for i in 1 2 3; do
  for c in 1 0; do
    echo "== SEG_GN_COOP=$c ($i)" >> $O/coop_ab.log; SEG_GN_COOP=$c timeout 200 $AB 2>/dev/null | grep -o "$G" >> $O/coop_ab.log
  done
done
cat $O/coop_ab.log
V=tools/experiments/libsegengine_diag.so
for i in 1 2; do
  for m in 2 0 1; do
    echo "== SEG_DIAG_W3_MODE=$m ($i)" >> $O/w3mode.log; SEGENGINE_LIB=$V SEG_DIAG_W3_MODE=$m timeout 200 $AB 2>/dev/null | grep -o "$G" >> $O/w3mode.log
  done
done
cat $O/w3mode.log
for c in 1 0; do SEG_GN_COOP=$c SEG_BENCH_ONLY=C4,C5,C2 timeout 300 python tools/bench_configs.py > $O/configs_coop$c.jsonl 2> $O/configs_coop$c.err; done
cut -c1-200 $O/configs_coop1.jsonl $O/configs_coop0.jsonl
rm -rf gpurun_out/trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --roofline-steps 0 --condition-seconds 0.2 --launch stream > $O/trace_run.log 2>&1
CSV=$(find gpurun_out/trace -name "*kernel_trace.csv" | head -1)
if [ -n "$CSV" ]; then python tools/trace_gaps.py $CSV > $O/trace_timeline.txt 2>&1; fi
rm -rf gpurun_out/trace
head -8 $O/trace_timeline.txt; grep -n "gn_bwd_coop" $O/trace_timeline.txt | head -30
