# round 3, session j: deep-level tilings inside the step after the epilogue change
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3j; mkdir -p $O
timeout 300 python tools/tune_conv3x.py --sets c3 --iters 30 > $O/tune_c3.jsonl 2> $O/tune.err
python - <<'PY'
import json
rows={}
for l in open('gpurun_out/r3j/tune_c3.jsonl'):
    try: d=json.loads(l)
    except: continue
    if 'us' in d: rows.setdefault(d['shape'],[]).append((d['us'], d['cfg']))
for k,v in rows.items(): print(k, sorted(v, key=lambda t: t[0])[:7])
PY
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --launch stream"
run() { tag=$1; shift; env "$@" timeout 200 $B > $O/$tag.json 2> $O/$tag.err; }
run base_a X=1
for c in 5 14 4 17; do run l24_$c SEG_C3X_MAP=64:64:24=$c; done
for c in 11 3; do run l12_$c SEG_C3X_MAP=128:128:12=$c; done
for c in 13 10; do run l6_$c SEG_C3X_MAP=256:256:6=$c; done
run base_b X=1
for f in $O/*.json; do echo "$f $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'])
except Exception as ex: print('ERR', ex)
")"; done
