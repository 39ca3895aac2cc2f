# round 4, session n: 32-output-channel workgroups (tilings 45 / 46) at the deepest level: step A/B, and the deepest shapes of C4 / C5
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4n; mkdir -p $O
run() { t=$1; shift
  env "$@" timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --launch stream > $O/$t.json 2> $O/$t.err
  python -c "
import json; d=json.loads(open('$O/$t.json').read().strip().splitlines()[-1]); print('$t', d['value'], d['ms_per_step'])"
}
{
run base_1 SEG_SUB_MB=0
run t46_6 SEG_C3X_MAP=256:256:6=46
run t45_6 SEG_C3X_MAP=256:256:6=45
run base_2 SEG_SUB_MB=0
run t46_6b SEG_C3X_MAP=256:256:6=46
run t45_6b SEG_C3X_MAP=256:256:6=45
} 2>&1 | tee $O/ab.log
timeout 400 python tools/tune_conv3x.py --sets c5,c4 --iters 20 > $O/tune.jsonl 2> $O/tune.err
python - <<'PY'
import json, collections
rows=[json.loads(l) for l in open('gpurun_out/r4n/tune.jsonl') if l.startswith('{') and '"us"' in l and '"best"' not in l]
by=collections.defaultdict(list)
for r in rows: by[r['shape']].append(r)
for k,v in by.items():
    v=sorted(v,key=lambda r:r['us'])
    print(k, [(r['cfg'], round(r['us'],1)) for r in v[:6]])
PY
