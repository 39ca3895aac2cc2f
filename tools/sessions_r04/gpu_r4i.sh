# round 4, session i: the other BASELINE configs with the new weight-gradient tiling (16-channel q-tiles, big box for every 16-channel gradient) vs the wide
# tile; bench.py other_configs leg after the warm-up change
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4i; mkdir -p $O
for arm in "SEG_W3_CQ=16" "SEG_W3_CQ=32" "SEG_W3_CQ=16 SEG_W3_BOX16=0" "SEG_W3_CQ=16 SEG_W3_TOTAL=512"; do
  echo "== $arm"; env $arm SEG_BENCH_ONLY=C2,C4,C5 timeout 200 python tools/bench_configs.py 2>/dev/null | cut -c1-130
done > $O/configs_ab.log 2>&1
cat $O/configs_ab.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])
for k,v in d['other_configs'].items(): print(k[:70], v.get('ms_per_step'), v.get('hbm_frac_of_fused_bound'), v.get('error'))"
timeout 600 python -m pytest tests/test_ops.py tests/test_fullsize.py -x -q -m gpu 2>&1 | tail -3
