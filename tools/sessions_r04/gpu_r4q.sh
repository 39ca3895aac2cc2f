# round 4, session q: 1x4 wave grids for the deep levels (every wave streams its own weight columns: no redundant L2 reads between the two M waves)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4q; mkdir -p $O
timeout 300 python tools/tune_conv3x.py --sets c3 --iters 30 > $O/tune.jsonl 2> $O/tune.err
python - <<'PY'
import json, collections
rows=[json.loads(l) for l in open('gpurun_out/r4q/tune.jsonl') if l.startswith('{') and '"us"' in l and '"best"' not in l]
by=collections.defaultdict(list)
for r in rows: by[r['shape']].append(r)
for k,v in by.items():
    v=sorted(v,key=lambda r:r['us'])
    print(k, [(r['cfg'], round(r['us'],1)) for r in v[:7]])
PY
run() { t=$1; shift
  env "$@" timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --launch stream > $O/$t.json 2> $O/$t.err
  python -c "
import json; d=json.loads(open('$O/$t.json').read().strip().splitlines()[-1]); print('$t', d['value'], d['ms_per_step'])"
}
{
run base_1 SEG_SUB_MB=0
run t49_12 SEG_C3X_MAP=128:128:12=49
run t50_12 SEG_C3X_MAP=128:128:12=50
run base_2 SEG_SUB_MB=0
run t49_12b SEG_C3X_MAP=128:128:12=49
run t50_12b SEG_C3X_MAP=128:128:12=50
} 2>&1 | tee $O/ab.log
