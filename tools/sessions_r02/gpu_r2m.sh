# round 2, GPU session M: workgroup count of the 16-channel weight gradient
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/r2m_ab.log
for cfg in "SEG_W3_TOTAL16=512" "SEG_W3_TOTAL16=1024" "SEG_W3_TOTAL16=2048" "SEG_W3_TOTAL16=1024 SEG_W3_MINBOX=4"; do
  echo "== $cfg" >> gpurun_out/r2m_ab.log
  env $cfg timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --roofline-steps 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*' >> gpurun_out/r2m_ab.log
  env $cfg timeout 120 python tools/bench_wgrad3.py child 2>/dev/null | tail -1 | cut -c1-120 >> gpurun_out/r2m_ab.log
done
cat gpurun_out/r2m_ab.log
