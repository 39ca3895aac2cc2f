# round 4, session m: deep-level halo convs (12^3 x 128, 6^3 x 256 channels): a 26-deep weight ring and 32-output-channel workgroups
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4m; mkdir -p $O
timeout 300 python tools/tune_conv3x.py --sets c3 --iters 30 > $O/tune.jsonl 2> $O/tune.err
python - <<'PY'
import json
rows=[json.loads(l) for l in open('gpurun_out/r4m/tune.jsonl') if l.startswith('{')]
for r in rows:
    if r.get('S') in (12, 6, 24) or True:
        pass
import collections
by=collections.defaultdict(list)
for r in rows:
    k=(r.get('shape') or (r.get('N'),r.get('S'),r.get('cin'),r.get('cout')))
    by[str(k)].append(r)
for k,v in by.items():
    v=sorted(v,key=lambda r:r.get('us',1e9))
    print(k, [(r.get('cfg'), round(r.get('us',0),1)) for r in v[:8]])
PY
tail -3 $O/tune.jsonl | cut -c1-300
run() { t=$1; shift
  env "$@" timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --launch stream > $O/$t.json 2> $O/$t.err
  python -c "
import json; d=json.loads(open('$O/$t.json').read().strip().splitlines()[-1]); print('$t', d['value'], d['ms_per_step'])"
}
{
run base_1 SEG_SUB_MB=0
run t44 SEG_C3X_MAP=128:128:12=44,256:256:6=44
run t45 SEG_C3X_MAP=128:128:12=45,256:256:6=45
run t46 SEG_C3X_MAP=128:128:12=46,256:256:6=46
run t45_6only SEG_C3X_MAP=256:256:6=45
run t45_12only SEG_C3X_MAP=128:128:12=45
run base_2 SEG_SUB_MB=0
} 2>&1 | tee $O/ab.log
