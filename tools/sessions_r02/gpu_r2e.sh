# round 2, GPU session E: stemx with hoisted index math, 32-bit GN index math, late release of weight-gradient batches
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_stemx.py -m gpu -x -q 2>&1 | tail -2 > gpurun_out/r2e_tests.log
cat gpurun_out/r2e_tests.log
rm -f gpurun_out/r2e_ab.log
for cfg in "SEG_FLUSH_LATE=0" "SEG_FLUSH_LATE=1" "SEG_FLUSH_LATE=1 SEG_FORK_BATCH=3" "SEG_FLUSH_LATE=1 SEG_FORK_BATCH=4" "SEG_FLUSH_LATE=1 SEG_STEMX_WGS=1024" "SEG_FLUSH_LATE=1 SEG_WGRAD3X=1"; do
  echo "== $cfg" >> gpurun_out/r2e_ab.log
  env $cfg timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --roofline-steps 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*' >> gpurun_out/r2e_ab.log
done
cat gpurun_out/r2e_ab.log
rm -rf gpurun_out/trace
timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --roofline-steps 0 > gpurun_out/trace_run.log 2>&1
T=$(find gpurun_out/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $T > gpurun_out/r2e_trace_gaps.txt 2>&1
rm -rf gpurun_out/trace
head -20 gpurun_out/r2e_trace_gaps.txt; grep -A 52 "per-phase" gpurun_out/r2e_trace_gaps.txt
