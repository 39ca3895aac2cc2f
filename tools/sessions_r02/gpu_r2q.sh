# round 2, GPU session Q: more runtime knobs of the launch / fence path
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/r2q_ab.log
for cfg in "SEG_X=0" "AMD_OPT_FLUSH=0" "AMD_OPT_FLUSH=1" "AMD_OPT_FLUSH=2" "AMD_OPT_FLUSH=3" "ROC_SYSTEM_SCOPE_SIGNAL=0" "DEBUG_HIP_KERNARG_COPY_OPT=0" "ROC_USE_FGS_KERNARG=0" "DEBUG_CLR_MAX_BATCH_SIZE=1" "SEG_X=1"; do
  echo "== $cfg" >> gpurun_out/r2q_ab.log
  env $cfg timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --roofline-steps 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*, .*"final_loss": [0-9.]*' | sed 's/"higher.*"final_loss"/"final_loss"/' >> gpurun_out/r2q_ab.log
done
cat gpurun_out/r2q_ab.log
