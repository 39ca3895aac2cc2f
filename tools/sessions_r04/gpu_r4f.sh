# round 4, session f: scheduling knobs re-swept on top of the pipelined weight-gradient kernels (one lease)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4f; mkdir -p $O
run() { # tag, env...
  t=$1; shift
  env "$@" timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --launch stream > $O/$t.json 2> $O/$t.err
  python -c "
import json; d=json.loads(open('$O/$t.json').read().strip().splitlines()[-1]); print('$t', d['value'], d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'])"
}
{
run base_1 SEG_SUB_MB=0
run hold2 SEG_HOLD_HEAVY_LVL=2
run hold3 SEG_HOLD_HEAVY_LVL=3
run hold2_32mb SEG_HOLD_HEAVY_LVL=2 SEG_HOLD_HEAVY_MB=32
run tail1 SEG_TAIL_WGRADS=1
run tail2 SEG_TAIL_WGRADS=2
run batch2 SEG_FORK_BATCH=2
run batch6 SEG_FORK_BATCH=6
run w3tot512 SEG_W3_TOTAL=512
run w3tot16_512 SEG_W3_TOTAL16=512
run wgcap16 SEG_WG_CAP=16777216
run wgcap16_tot4096 SEG_WG_CAP=16777216 SEG_WG_TOTAL=4096
run noside SEG_WGRAD_STREAM=0
run noside_w3tot1024 SEG_WGRAD_STREAM=0 SEG_W3_TOTAL=1024 SEG_W3_TOTAL16=1024 SEG_WG_CAP=16777216 SEG_WG_TOTAL=4096
run prio0 SEG_SIDE_PRIO=0
run base_2 SEG_SUB_MB=0
} 2>&1 | tee $O/ab.log
