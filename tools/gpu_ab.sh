# A/B of environment knobs on one MI355X (un-bracketed bench, 30 steps each): bash tools/gpu_ab.sh "SEG_FORK_BATCH=6" "SEG_FORK_BATCH=8" ...
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/ab.log
for cfg in "$@"; do
  echo "== $cfg" >> gpurun_out/ab.log
  env $cfg timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --roofline-steps 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*' >> gpurun_out/ab.log
done
cat gpurun_out/ab.log
