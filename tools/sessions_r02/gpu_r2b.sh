# round 2, GPU session B: wgrad3x on the GPU (bit-exact), op-level old vs new, step-level A/B of the policy knobs, kernel stats
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_ops.py tests/test_conv3x.py -m gpu -x -q -k "wgrad3 or conv3x" 2>&1 | tail -3 > gpurun_out/r2b_ops_tests.log
cat gpurun_out/r2b_ops_tests.log
timeout 200 python tools/bench_wgrad3.py SEG_WGRAD3X=0 SEG_WGRAD3X=1 SEG_WGRAD3X=1,SEG_W3X_TOTAL=512 SEG_WGRAD3X=1,SEG_W3X_MINBOX=8 > gpurun_out/r2b_wgrad3_ops.log 2>&1
cat gpurun_out/r2b_wgrad3_ops.log
rm -f gpurun_out/r2b_ab.log
for cfg in "SEG_WGRAD3X=0" "SEG_WGRAD3X=1" "SEG_W3X_TOTAL=512" "SEG_W3X_MINBOX=2" "SEG_W3X_MINBOX=8" "SEG_WGRAD_STREAM=0"; do
  echo "== $cfg" >> gpurun_out/r2b_ab.log
  env $cfg timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --roofline-steps 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*' >> gpurun_out/r2b_ab.log
done
cat gpurun_out/r2b_ab.log
rm -rf gpurun_out/prof
timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o step -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --roofline-steps 0 > gpurun_out/prof_run.log 2>&1
DB=$(find gpurun_out/prof -name "*.db" | head -1)
if [ -n "$DB" ]; then python profiles/summarize_rocpd.py $DB 50 > gpurun_out/r2b_kernel_stats.txt 2>&1; fi
rm -rf gpurun_out/prof
head -52 gpurun_out/r2b_kernel_stats.txt
