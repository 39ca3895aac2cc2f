# round 5, session i: the persistent halo conv (tiling 51: next halo copied under the epilogue) - bit-exact GPU tests, standalone against tilings 17 / 14 / 3,
# in-call A/B of the driver's command with SEG_C3X_MAP selecting it for the 48^3 x 32-channel level
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r5i; mkdir -p $O
timeout 600 python -m pytest tests/test_conv3x.py -m gpu -x -q > $O/conv3x_tests.log 2>&1; tail -2 $O/conv3x_tests.log
timeout 300 python tools/bench_conv3x_cfgs.py > $O/standalone.jsonl 2> $O/standalone.err; cat $O/standalone.jsonl
DRV="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs"
for arm in "" "32:32:48=51" "" "32:32:48=51"; do
  echo "== SEG_C3X_MAP=$arm" >> $O/ab.log
  SEG_C3X_MAP=$arm timeout 300 $DRV 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:(v['ms_per_step'],v['frac']) for k,v in d['kernel_families'].items() if 'halo' in k})" >> $O/ab.log
done
cat $O/ab.log
