# round 4, session o: 12^3 level with 4x4x12 boxes and 32-channel workgroups (tilings 47 / 48); new default (tiling 45 at the deepest level) vs before
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4o; mkdir -p $O
timeout 300 python tools/tune_conv3x.py --sets c3 --iters 30 > $O/tune.jsonl 2> $O/tune.err
python - <<'PY'
import json, collections
rows=[json.loads(l) for l in open('gpurun_out/r4o/tune.jsonl') if l.startswith('{') and '"us"' in l and '"best"' not in l]
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
run new_1 SEG_SUB_MB=0
run old_1 SEG_C3X_MAP=256:256:6=7
run t47_12 SEG_C3X_MAP=128:128:12=47
run t48_12 SEG_C3X_MAP=128:128:12=48
run new_2 SEG_SUB_MB=0
run old_2 SEG_C3X_MAP=256:256:6=7
} 2>&1 | tee $O/ab.log
