# round 3, session g: batched y loads in the dgrad-reduce epilogue + hoisted parameter loads in the fold prologues
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3g; mkdir -p $O
timeout 600 python -m pytest tests/test_engine.py tests/test_conv3x.py -m gpu -x -q 2>&1 | tail -3 > $O/tests.log; cat $O/tests.log
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --launch stream"
run() { tag=$1; shift; env "$@" timeout 200 $B > $O/$tag.json 2> $O/$tag.err; }
run new_a X=1
run norfuse_a SEG_GN_RFUSE=0
run new_b X=1
run norfuse_b SEG_GN_RFUSE=0
run nofold SEG_GN_FOLD=0
run forkb2 SEG_FORK_BATCH=2
run forkb6 SEG_FORK_BATCH=6
run heavy8 SEG_FORK_HEAVY_MB=8
run heavy64 SEG_FORK_HEAVY_MB=64
for f in $O/*.json; do echo "$f $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], 'host', l.get('host_enqueue_ms_per_step'), 'roof', (l.get('roofline') or {}).get('frac'), 'mfma_us', (l.get('roofline_mfma') or {}).get('avg_launch_us'))
except Exception as ex: print('ERR', ex)
")"; done
rm -rf gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o step -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --roofline-steps 0 --condition-seconds 0.2 --launch stream > $O/prof_run.log 2>&1
DB=$(find gpurun_out/prof -name "*.db" | head -1)
if [ -n "$DB" ]; then python profiles/summarize_rocpd.py $DB 50 > $O/kernel_stats.txt 2>&1; fi
rm -rf gpurun_out/prof
head -40 $O/kernel_stats.txt
