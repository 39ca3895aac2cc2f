# round 3, session l: partial-tile policy of the weight-gradient kernels (fewer, larger slices -> less partial-tile traffic on the co-critical side stream)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3l; mkdir -p $O
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --launch stream"
run() { tag=$1; shift; env "$@" timeout 200 $B > $O/$tag.json 2> $O/$tag.err; }
run base_a X=1
run minbox9 SEG_W3_MINBOX=9
run minbox12 SEG_W3_MINBOX=12
run minbox4 SEG_W3_MINBOX=4
run total384 SEG_W3_TOTAL=384
run total256 SEG_W3_TOTAL=256
run total768 SEG_W3_TOTAL=768
run t16_512 SEG_W3_TOTAL16=512
run t16_2048 SEG_W3_TOTAL16=2048
run wg1024 SEG_WG_TOTAL=1024
run wg4096 SEG_WG_TOTAL=4096
run wgcap1 SEG_WG_CAP=1048576
run base_b X=1
for f in $O/*.json; do echo "$f $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'])
except Exception as ex: print('ERR', ex)
")"; done
