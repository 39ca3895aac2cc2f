# round 3, session h: conv3x epilogue that stores straight from the accumulators (swapped MFMA operands) against the previous binary
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3h; mkdir -p $O
timeout 600 python -m pytest tests/test_conv3x.py tests/test_engine.py -m gpu -x -q 2>&1 | tail -3 > $O/tests.log; cat $O/tests.log
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --launch stream"
PREV=/root/repo/pytorchdeeplearing_amd/lib/variants/libsegengine_prev.so
run() { tag=$1; shift; env "$@" timeout 200 $B > $O/$tag.json 2> $O/$tag.err; }
run new_a X=1
run prev_a SEGENGINE_LIB=$PREV
run new_b X=1
run prev_b SEGENGINE_LIB=$PREV
run new_c X=1
for f in $O/*.json; do echo "$f $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], 'host', l.get('host_enqueue_ms_per_step'), 'roof', (l.get('roofline') or {}).get('frac'), 'mfma_us', (l.get('roofline_mfma') or {}).get('avg_launch_us'))
except Exception as ex: print('ERR', ex)
")"; done
timeout 300 python tools/tune_conv3x.py --sets top --iters 30 > $O/tune_top.jsonl 2> $O/tune.err
python - <<'PY'
import json
for l in open('gpurun_out/r3h/tune_top.jsonl'):
    try: d=json.loads(l)
    except: continue
    if 'us' in d: print(d['shape'], d['cfg'], d['us'])
PY
