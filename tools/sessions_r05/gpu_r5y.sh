# round 5, session y: gn_bwd_group_kernel (one-launch GroupNorm backward of the 6^3 level) with every later phase's memory reads requested in front of the first
# pass and the thread's item kept in registers for the second pass - engine parity tests on the GPU, in-call A/B against the library linked with the previous norm.hip
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r5y; mkdir -p $O
timeout 900 python -m pytest tests/test_engine.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
DRV="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --roofline-steps 5"
SHOW='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], {k:(v["ms_per_step"],v["launches_per_step"]) for k,v in d["kernel_families"].items() if k=="gn_small"})'
for i in 1 2 3 4; do
  echo -n "previous norm.hip: " >> $O/ab.log; SEGENGINE_LIB=$PWD/pytorchdeeplearing_amd/lib/ab/libsegengine_prevnorm.so timeout 300 $DRV 2>/dev/null | python -c "$SHOW" >> $O/ab.log
  echo -n "current: " >> $O/ab.log; timeout 300 $DRV 2>/dev/null | python -c "$SHOW" >> $O/ab.log
done
cat $O/ab.log
