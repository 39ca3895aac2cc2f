# round 3, session b: virtual activations (GN + dropout + ReLU applied by the consumer conv) on the GPU; CU-mask A/B of the weight-gradient stream
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3b; mkdir -p $O
timeout 900 python -m pytest tests/test_engine.py tests/test_boundary.py tests/test_conv3x.py -m gpu -x -q 2>&1 | tail -5 > $O/tests.log; cat $O/tests.log
B="python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline"
run() { tag=$1; shift; env "$@" timeout 200 $B > $O/$tag.json 2> $O/$tag.err; }
run vact1 SEG_GN_VACT=1
run vact0 SEG_GN_VACT=0
run vact1_b SEG_GN_VACT=1
run vact0_b SEG_GN_VACT=0
run cus128 SEG_SIDE_CUS=128
run cus128s2 SEG_SIDE_CUS=128 SEG_SIDE_CU_STRIDE=2
run cus64 SEG_SIDE_CUS=64
run cus64s4 SEG_SIDE_CUS=64 SEG_SIDE_CU_STRIDE=4
run cus192 SEG_SIDE_CUS=192
run cus96s8 SEG_SIDE_CUS=96 SEG_SIDE_CU_STRIDE=8
run prio0 SEG_SIDE_PRIO=0
for f in $O/*.json; do echo "$f $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], 'host', l.get('host_enqueue_ms_per_step'), 'roof', (l.get('roofline') or {}).get('kernel','')[:30], (l.get('roofline') or {}).get('frac'))
except Exception as ex: print('ERR', ex)
")"; done
# kernel trace of the new default
rm -rf gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o step -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --roofline-steps 0 --condition-seconds 0.2 > $O/prof_run.log 2>&1
DB=$(find gpurun_out/prof -name "*.db" | head -1)
if [ -n "$DB" ]; then python profiles/summarize_rocpd.py $DB 50 > $O/kernel_stats.txt 2>&1; fi
CSV=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1)
if [ -n "$CSV" ]; then python tools/trace_gaps.py $CSV > $O/trace_timeline.txt 2>&1; fi
rm -rf gpurun_out/prof
head -30 $O/kernel_stats.txt
