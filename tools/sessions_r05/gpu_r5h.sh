# round 5, session h: conv3x epilogue with DPP row sums + LDS-staged bias - phase trace after, bit-exact GPU tests, in-call A/B of the driver's command against the
# previous binary (lib/variants/libsegengine_prev.so)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r5h; mkdir -p $O
timeout 200 python tools/trace_conv3x.py 2> $O/conv3x_phase_trace.log > /dev/null; grep "conv3x trace" $O/conv3x_phase_trace.log | awk 'NR%3==0'
timeout 600 python -m pytest tests/test_conv3x.py tests/test_ops.py -m gpu -x -q > $O/ops_tests.log 2>&1; tail -2 $O/ops_tests.log
DRV="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs"
for arm in prev new prev new; do
  if [ $arm = prev ]; then export SEGENGINE_LIB=$PWD/pytorchdeeplearing_amd/lib/variants/libsegengine_prev.so; else unset SEGENGINE_LIB; fi
  echo "== $arm" >> $O/ab.log
  timeout 300 $DRV 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:(v['ms_per_step'],v['frac']) for k,v in d['kernel_families'].items() if 'halo' in k})" >> $O/ab.log
done
unset SEGENGINE_LIB
cat $O/ab.log
