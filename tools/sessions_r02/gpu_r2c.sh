# round 2, GPU session C: fused input block + Cin=16 conv3x + XCD-aware box order: op tests, step A/B, kernel stats
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_stemx.py tests/test_conv3x.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r2c_ops_tests.log
cat gpurun_out/r2c_ops_tests.log
rm -f gpurun_out/r2c_ab.log
for cfg in "SEG_STEMX=0 SEG_C3X_REMAP=0 SEG_C3X_CFG=-1" "SEG_STEMX=1 SEG_C3X_REMAP=0" "SEG_STEMX=1 SEG_C3X_REMAP=1" "SEG_STEMX=1 SEG_C3X_REMAP=1 SEG_C3X_MAP=16:16:96=25" "SEG_STEMX=1 SEG_STEMX_WGS=1024" "SEG_STEMX=1 SEG_STEMX_WGS=4096"; do
  echo "== $cfg" >> gpurun_out/r2c_ab.log
  env $cfg timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --roofline-steps 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*' >> gpurun_out/r2c_ab.log
done
cat gpurun_out/r2c_ab.log
rm -rf gpurun_out/prof
timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o step -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --roofline-steps 0 > gpurun_out/prof_run.log 2>&1
DB=$(find gpurun_out/prof -name "*.db" | head -1)
if [ -n "$DB" ]; then python profiles/summarize_rocpd.py $DB 50 > gpurun_out/r2c_kernel_stats.txt 2>&1; fi
rm -rf gpurun_out/prof
head -40 gpurun_out/r2c_kernel_stats.txt
timeout 400 python -m pytest tests/test_engine.py tests/test_ops.py -m gpu -x -q 2>&1 | tail -3
