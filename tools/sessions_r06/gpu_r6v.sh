# previous binary (a0d25bddb709, sources of c20c064) against the final binary in ONE call
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r6v; mkdir -p $O
P=pytorchdeeplearing_amd/lib/variants/libsegengine_prev.so
AB="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --roofline-steps 0"
G='"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*'
SEGENGINE_LIB=$P python -c "from pytorchdeeplearing_amd import _capi; print(_capi.product_library().build_info())" >> $O/ab.log 2>&1
python -c "from pytorchdeeplearing_amd import _capi; print(_capi.product_library().build_info())" >> $O/ab.log 2>&1
for i in 1 2 3 4; do
  echo "== previous binary ($i)" >> $O/ab.log; env SEGENGINE_LIB=$P timeout 200 $AB 2>/dev/null | grep -o "$G" >> $O/ab.log
  env SEGENGINE_LIB=$P SEG_BENCH_ONLY=C4,C5 SEG_BENCH_NOPROF=1 timeout 200 python tools/bench_configs.py 2>/dev/null | cut -c1-90 >> $O/ab.log
  echo "== final binary ($i)" >> $O/ab.log; timeout 200 $AB 2>/dev/null | grep -o "$G" >> $O/ab.log
  env SEG_BENCH_ONLY=C4,C5 SEG_BENCH_NOPROF=1 timeout 200 python tools/bench_configs.py 2>/dev/null | cut -c1-90 >> $O/ab.log
  echo "== final binary, SEG_C3X16_REUSE=0 ($i)" >> $O/ab.log; env SEG_C3X16_REUSE=0 timeout 200 $AB 2>/dev/null | grep -o "$G" >> $O/ab.log
done
cat $O/ab.log
# workgroups of the 16-channel 96^3 weight gradient (diagnostic build of the final sources, -DSEG_DIAG conv3.hip: SEG_W3_TOTAL16)
V=pytorchdeeplearing_amd/lib/variants/libsegengine_w3t16.so
for i in 1 2 3; do
for cfg in "SEG_W3_TOTAL16=256" "SEG_W3_TOTAL16=384" "SEG_W3_TOTAL16=512" "SEG_W3_TOTAL16=768"; do
  echo "== $cfg ($i)" >> $O/w3t16.log; env SEGENGINE_LIB=$V $cfg timeout 200 $AB 2>/dev/null | grep -o "$G" >> $O/w3t16.log
done; done
cat $O/w3t16.log
