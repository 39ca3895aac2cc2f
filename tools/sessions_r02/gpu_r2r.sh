# round 2, GPU session R: last single-knob A/B on the 96^3-level kernels
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/r2r_ab.log
for cfg in "SEG_X=0" "SEG_C3X_MAP=16:16:96=25" "SEG_STEMX_WGS=1024" "SEG_STEMX_WGS=4096" "SEG_PACK_WGS=512" "SEG_FOLD_WGS=3072" "SEG_X=1"; do
  echo "== $cfg" >> gpurun_out/r2r_ab.log
  env $cfg timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --roofline-steps 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*' >> gpurun_out/r2r_ab.log
done
cat gpurun_out/r2r_ab.log
