# round 2, GPU session G: early release of heavy weight gradients; full-size oracle tests
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/r2g_ab.log
for cfg in "SEG_FORK_HEAVY_MB=1000000" "SEG_FORK_HEAVY_MB=100" "SEG_FORK_HEAVY_MB=16" "SEG_FORK_HEAVY_MB=4" "SEG_FORK_HEAVY_MB=16 SEG_FLUSH_LATE=0" "SEG_FORK_HEAVY_MB=16 SEG_FORK_BATCH=3" "SEG_FORK_HEAVY_MB=0"; do
  echo "== $cfg" >> gpurun_out/r2g_ab.log
  env $cfg timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --roofline-steps 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*' >> gpurun_out/r2g_ab.log
done
cat gpurun_out/r2g_ab.log
rm -rf gpurun_out/trace
timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --roofline-steps 0 > gpurun_out/trace_run.log 2>&1
T=$(find gpurun_out/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $T > gpurun_out/r2g_trace_timeline.txt 2>&1
rm -rf gpurun_out/trace
head -12 gpurun_out/r2g_trace_timeline.txt
timeout 900 python -m pytest tests/test_fullsize.py -m gpu -x -q -s 2>&1 | tail -30 > gpurun_out/r2g_fullsize.log
cat gpurun_out/r2g_fullsize.log
