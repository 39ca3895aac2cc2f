# round 5, session l: persistent Cin == 16 halo conv (tilings 52 / 53: next halo copied under the epilogue) - bit-exact GPU tests, standalone against tiling 25 / 24 on the
# finest-level shapes of C3 / C4 / C5, in-call A/B of the driver's command with SEG_C3X_MAP selecting it for the 96^3 level
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r5l; mkdir -p $O
timeout 600 python -m pytest tests/test_conv3x.py -m gpu -x -q > $O/conv3x_tests.log 2>&1; tail -2 $O/conv3x_tests.log
timeout 300 python tools/bench_conv3x_cfgs.py "4,96,16,16:25,52,53,24" "2,128,16,16:25,52,53" "1,160,16,16:25,52,53" > $O/standalone.jsonl 2> $O/standalone.err; cat $O/standalone.jsonl; tail -2 $O/standalone.err
for w in 512 1024; do echo "SEG_C3Q16_WGS=$w" >> $O/standalone_wgs.jsonl; SEG_C3Q16_WGS=$w timeout 200 python tools/bench_conv3x_cfgs.py "4,96,16,16:52" >> $O/standalone_wgs.jsonl 2>/dev/null; done; cat $O/standalone_wgs.jsonl
DRV="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs"
for arm in "" "16:16:96=52" "" "16:16:96=52"; do
  echo "== SEG_C3X_MAP=$arm" >> $O/ab.log
  SEG_C3X_MAP=$arm timeout 300 $DRV 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:(v['ms_per_step'],v['frac']) for k,v in d['kernel_families'].items() if 'halo' in k})" >> $O/ab.log
done
cat $O/ab.log
