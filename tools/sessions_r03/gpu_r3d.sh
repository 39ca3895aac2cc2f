# round 3, session d: the train step as a HIP graph (capture + replay) against the stream launches; what the GN statistics epilogue costs; loss curves
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3d; mkdir -p $O
timeout 300 python -m pytest tests/test_engine.py -m gpu -x -q -k "graph_replay" 2>&1 | tail -12 > $O/tests_graph.log; cat $O/tests_graph.log
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
run() { tag=$1; shift; env "$@" timeout 200 $B > $O/$tag.json 2> $O/$tag.err; }
run auto X=1
env X=1 timeout 200 $B --launch stream > $O/stream.json 2> $O/stream.err
env X=1 timeout 200 $B --launch graph > $O/graph.json 2> $O/graph.err
env X=1 timeout 200 $B --launch graph --steps 100 > $O/graph100.json 2> $O/graph100.err
env X=1 timeout 200 $B --launch stream --steps 100 > $O/stream100.json 2> $O/stream100.err
# a slow host, emulated: the enqueueing thread shares ONE core with seven busy loops
for i in 1 2 3 4 5 6 7; do taskset -c 3 python -c "while True: pass" & done
sleep 1
taskset -c 3 env X=1 timeout 200 $B --launch stream > $O/slow_stream.json 2> $O/slow_stream.err
taskset -c 3 env X=1 timeout 200 $B --launch graph > $O/slow_graph.json 2> $O/slow_graph.err
taskset -c 3 env X=1 timeout 200 $B --launch auto > $O/slow_auto.json 2> $O/slow_auto.err
kill %1 %2 %3 %4 %5 %6 %7 2>/dev/null
run nostats SEG_DIAG_NOSTATS=1
run norfuse SEG_GN_RFUSE=0
for f in $O/*.json; do echo "$f $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], 'host', l.get('host_enqueue_ms_per_step'), l.get('launch_mode'), l.get('launch_probe_ms_per_step'), 'mfma_us', (l.get('roofline_mfma') or {}).get('avg_launch_us'))
except Exception as ex: print('ERR', ex)
")"; done
tail -3 $O/graph.err
SEG_FULLSIZE_REPORT=$O/loss_curves.txt timeout 600 python -m pytest tests/test_fullsize.py -m gpu -x -q -k "thirty_step" -s 2>&1 | grep -E "loss curve|passed|failed|Error" | head; cat $O/loss_curves.txt
timeout 900 python -m pytest tests/test_wrappers.py tests/test_boundary.py -m gpu -x -q 2>&1 | tail -4
