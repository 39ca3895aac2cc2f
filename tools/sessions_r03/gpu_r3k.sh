# round 3, session k: hold the heavy (96^3 / 48^3) decoder weight gradients until the backward pass reaches the latency-bound deep levels
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3k; mkdir -p $O
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --launch stream"
run() { tag=$1; shift; env "$@" timeout 200 $B > $O/$tag.json 2> $O/$tag.err; }
run base_a X=1
run hold2_64 SEG_HOLD_HEAVY_LVL=2 SEG_HOLD_HEAVY_MB=64
run hold2_16 SEG_HOLD_HEAVY_LVL=2 SEG_HOLD_HEAVY_MB=16
run hold3_64 SEG_HOLD_HEAVY_LVL=3 SEG_HOLD_HEAVY_MB=64
run hold3_16 SEG_HOLD_HEAVY_LVL=3 SEG_HOLD_HEAVY_MB=16
run hold1_64 SEG_HOLD_HEAVY_LVL=1 SEG_HOLD_HEAVY_MB=64
run hold4_16 SEG_HOLD_HEAVY_LVL=4 SEG_HOLD_HEAVY_MB=16
run base_b X=1
run hold2_64_b SEG_HOLD_HEAVY_LVL=2 SEG_HOLD_HEAVY_MB=64
for f in $O/*.json; do echo "$f $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], 'roof', (l.get('roofline') or {}).get('frac'), 'apply_us', (l.get('roofline') or {}).get('avg_launch_us'))
except Exception as ex: print('ERR', ex)
")"; done
timeout 300 python -m pytest tests/test_engine.py -m gpu -x -q -k "parity or three_steps or backward_op" 2>&1 | tail -2
