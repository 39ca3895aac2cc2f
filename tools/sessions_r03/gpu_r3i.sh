# round 3, session i: tiling choices after the epilogue change (standalone winners inside the step), higher-occupancy variants
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3i; mkdir -p $O
timeout 300 python tools/tune_conv3x.py --sets top --iters 30 > $O/tune_top.jsonl 2> $O/tune.err
python - <<'PY'
import json
for l in open('gpurun_out/r3i/tune_top.jsonl'):
    try: d=json.loads(l)
    except: continue
    if 'us' in d: print(d['shape'], d['cfg'], d['us'], d.get('same_as_conv3'))
PY
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --launch stream"
run() { tag=$1; shift; env "$@" timeout 200 $B > $O/$tag.json 2> $O/$tag.err; }
run base_a X=1
run c16_25 SEG_C3X_MAP=16:16:96=25
run c32_17 SEG_C3X_MAP=32:32:48=17
run c32_20 SEG_C3X_MAP=32:32:48=20
run c32_23 SEG_C3X_MAP=32:32:48=23
run c32_21 SEG_C3X_MAP=32:32:48=21
run c32_22 SEG_C3X_MAP=32:32:48=22
run base_b X=1
run both SEG_C3X_MAP=16:16:96=25,32:32:48=17
for f in $O/*.json; do echo "$f $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], 'mfma_us', (l.get('roofline_mfma') or {}).get('avg_launch_us'))
except Exception as ex: print('ERR', ex)
")"; done
