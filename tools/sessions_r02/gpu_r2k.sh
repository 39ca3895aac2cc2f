# round 2, GPU session K: backward-only weight layouts packed on the side stream; stream / tiling A/B inside the step
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_engine.py -m gpu -x -q -k "f32 or train_steps or bucketed or lowp" 2>&1 | tail -2 > gpurun_out/r2k_tests.log
cat gpurun_out/r2k_tests.log
rm -f gpurun_out/r2k_ab.log
for cfg in "SEG_PACK_SPLIT=0" "SEG_PACK_SPLIT=1" "SEG_WGRAD_STREAM=0" "SEG_SIDE_PRIO=0" "SEG_C3X_MAP=128:128:12=3" "SEG_C3X_MAP=128:128:12=11" "SEG_C3X_MAP=64:64:24=3" "SEG_C3X_MAP=64:64:24=14" "SEG_C3X_MAP=64:64:24=11" "SEG_C3X_MAP=256:256:6=13" "SEG_C3X_MAP=32:32:48=17" "SEG_C3X_MAP=32:32:48=3"; do
  echo "== $cfg" >> gpurun_out/r2k_ab.log
  env $cfg timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --roofline-steps 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*' >> gpurun_out/r2k_ab.log
done
cat gpurun_out/r2k_ab.log
