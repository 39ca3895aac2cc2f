# round 2, GPU session D: stemx pipeline, virtual head gradient, vector coefficient loads: tests, A/B, per-queue trace
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_stemx.py tests/test_engine.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r2d_tests.log
cat gpurun_out/r2d_tests.log
rm -f gpurun_out/r2d_ab.log
for cfg in "SEG_VHEAD=0" "SEG_VHEAD=1" "SEG_VHEAD=1 SEG_HEAD_BWD_WGS=512" "SEG_VHEAD=1 SEG_HEAD_BWD_WGS=4096" "SEG_VHEAD=1 SEG_FORK_BATCH=3" "SEG_VHEAD=1 SEG_FORK_BATCH=12"; do
  echo "== $cfg" >> gpurun_out/r2d_ab.log
  env $cfg timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --roofline-steps 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*' >> gpurun_out/r2d_ab.log
done
cat gpurun_out/r2d_ab.log
rm -rf gpurun_out/trace
timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --roofline-steps 0 > gpurun_out/trace_run.log 2>&1
T=$(find gpurun_out/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $T > gpurun_out/r2d_trace_gaps.txt 2>&1
rm -rf gpurun_out/trace
head -60 gpurun_out/r2d_trace_gaps.txt; grep -A 16 "per-phase" gpurun_out/r2d_trace_gaps.txt | head -70
