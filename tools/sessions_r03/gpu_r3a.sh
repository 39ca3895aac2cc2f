# round 3, session a: is the 20/5 vs 50/10 gap of round 2 reproducible on ONE box, and what does the new bench / one-call step give?
cd /root/repo; mkdir -p gpurun_out  # (run as: gpurun -- bash tools/sessions_r03/<this file>); export TMPDIR=/tmp
O=gpurun_out/r3a; mkdir -p $O
timeout 600 python -m pytest tests/test_engine.py tests/test_boundary.py -m gpu -x -q 2>&1 | tail -4 > $O/tests.log; cat $O/tests.log
# legacy bench (round-2 file, same library): driver's command vs builder's command, twice each, fresh processes
for i in 1 2; do
  timeout 200 python tools/bench_r02_legacy.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/legacy_20_5_$i.json 2> $O/legacy_20_5_$i.err
  timeout 200 python tools/bench_r02_legacy.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline > $O/legacy_50_10_$i.json 2> $O/legacy_50_10_$i.err
done
# new bench, the driver's exact command, three fresh processes; then without conditioning; then the multi-call host path
for i in 1 2 3; do
  timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/new_20_5_$i.json 2> $O/new_20_5_$i.err
done
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --condition-seconds 0 > $O/new_20_5_nocond.json 2> $O/new_nocond.err
SEG_ONE_CALL=0 timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/new_20_5_multicall.json 2> $O/new_multicall.err
timeout 200 python bench.py --gpus 1 --steps 200 --warmup 5 --no-cpu-baseline > $O/new_200_5.json 2> $O/new_200.err
# the full driver line (with the baselines) once
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/new_full.json 2> $O/new_full.err
for f in $O/*.json; do echo "$f $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], 'host', l.get('host_enqueue_ms_per_step'), 'cond', l.get('conditioning_steps'), 'roof', (l.get('roofline') or {}).get('frac'))
except Exception as ex: print('ERR', ex)
")"; done
rocm-smi --showclocks 2>/dev/null | head -20
