# round 2, GPU session J: replica counts by level for the folded finalize
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_engine.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r2j_tests.log
cat gpurun_out/r2j_tests.log
rm -f gpurun_out/r2j_ab.log
for cfg in "SEG_GN_FOLD=0" "SEG_GN_FOLD=1" "SEG_FOLD_WGS=1024" "SEG_GN_FOLD=1 SEG_FORK_BATCH=2" "SEG_GN_FOLD=1 SEG_FORK_BATCH=4"; do
  echo "== $cfg" >> gpurun_out/r2j_ab.log
  env $cfg timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --roofline-steps 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*' >> gpurun_out/r2j_ab.log
done
cat gpurun_out/r2j_ab.log
rm -rf gpurun_out/trace
timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --roofline-steps 0 > gpurun_out/trace_run.log 2>&1
T=$(find gpurun_out/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $T > gpurun_out/r2j_trace_timeline.txt 2>&1
rm -rf gpurun_out/trace
head -5 gpurun_out/r2j_trace_timeline.txt; grep -A 40 "per-phase" gpurun_out/r2j_trace_timeline.txt | grep "==\|gn_\|igemm"
