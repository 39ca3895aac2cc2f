# round 5, session x: four reduction slices per stage in conv_igemm_kernel for launches of <= 128 workgroups (the stride-2 / transposed convs of the 12^3 and
# 6^3 levels) - operator / engine parity tests on the GPU, in-call A/B of the driver's command against the library linked with the previous conv.hip, the other configs
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r5x; mkdir -p $O
timeout 900 python -m pytest tests/test_ops.py tests/test_engine.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
DRV="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --roofline-steps 5"
SHOW='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], {k:(v["ms_per_step"],v["launches_per_step"]) for k,v in d["kernel_families"].items() if k=="generic_conv"})'
for i in 1 2 3 4; do
  echo -n "previous conv.hip: " >> $O/ab.log; SEGENGINE_LIB=$PWD/pytorchdeeplearing_amd/lib/ab/libsegengine_prevconv.so timeout 300 $DRV 2>/dev/null | python -c "$SHOW" >> $O/ab.log
  echo -n "current: " >> $O/ab.log; timeout 300 $DRV 2>/dev/null | python -c "$SHOW" >> $O/ab.log
done
cat $O/ab.log
for i in 1 2; do
  echo "== previous conv.hip" >> $O/configs.log; SEGENGINE_LIB=$PWD/pytorchdeeplearing_amd/lib/ab/libsegengine_prevconv.so SEG_BENCH_ONLY=C2,C4,C5 timeout 300 python tools/bench_configs.py 2>/dev/null | cut -c1-100 >> $O/configs.log
  echo "== current" >> $O/configs.log; SEG_BENCH_ONLY=C2,C4,C5 timeout 300 python tools/bench_configs.py 2>/dev/null | cut -c1-100 >> $O/configs.log
done
cat $O/configs.log
