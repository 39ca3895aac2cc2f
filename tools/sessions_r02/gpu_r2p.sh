# round 2, GPU session P: runtime launch-path knobs
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/r2p_ab.log
for cfg in "SEG_X=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "GPU_MAX_HW_QUEUES=2" "GPU_MAX_HW_QUEUES=8" "HSA_ENABLE_INTERRUPT=0" "SEG_X=1"; do
  echo "== $cfg" >> gpurun_out/r2p_ab.log
  env $cfg timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --roofline-steps 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*' >> gpurun_out/r2p_ab.log
done
cat gpurun_out/r2p_ab.log
