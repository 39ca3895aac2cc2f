# round 3, session f: statistics from the accumulators (conv3x epilogue) A/B; the graph path again (own stream); entry-script test
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3f; mkdir -p $O
timeout 600 python -m pytest tests/test_conv3x.py tests/test_engine.py -m gpu -x -q 2>&1 | tail -6 > $O/tests.log; cat $O/tests.log
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
run() { tag=$1; shift; env "$@" timeout 200 $B > $O/$tag.json 2> $O/$tag.err; }
run auto_a X=1
run auto_b X=1
env X=1 timeout 200 $B --launch stream > $O/stream.json 2> $O/stream.err
env X=1 timeout 200 $B --launch graph > $O/graph.json 2> $O/graph.err
run nostats SEG_DIAG_NOSTATS=1
for i in 1 2 3 4 5 6 7; do taskset -c 3 python -c "while True: pass" & done
sleep 1
taskset -c 3 env X=1 timeout 200 $B --launch stream > $O/slow_stream.json 2> $O/slow_stream.err
taskset -c 3 env X=1 timeout 200 $B --launch graph > $O/slow_graph.json 2> $O/slow_graph.err
taskset -c 3 env X=1 timeout 200 $B --launch auto > $O/slow_auto.json 2> $O/slow_auto.err
kill %1 %2 %3 %4 %5 %6 %7 2>/dev/null
for f in $O/*.json; do echo "$f $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], 'host', l.get('host_enqueue_ms_per_step'), l.get('launch_mode'), l.get('launch_probe_ms_per_step'), 'mfma_us', (l.get('roofline_mfma') or {}).get('avg_launch_us'))
except Exception as ex: print('ERR', ex)
")"; done
timeout 600 python -m pytest tests/test_wrappers.py tests/test_boundary.py -m gpu -x -q 2>&1 | tail -5
