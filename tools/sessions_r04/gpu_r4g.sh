# round 4, session g: (1) the new bench.py line (families, other_configs), (2) N3 end to end (tools/bench_pipeline.py), (3) GPU tests touched this round
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4g; mkdir -p $O
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err; tail -c 600 $O/bench_full.err
python -c "
import json; d=json.loads(open('$O/bench_full.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'timed_region_s', d.get('timed_region_s'))
for k in ('roofline','roofline_2','roofline_3'):
    r=d.get(k) or {}; print(k, r.get('kernel','')[:40], r.get('bound'), r.get('frac'), r.get('frac_minus_bracket'), 'ms/step', r.get('ms_per_step'), 'traffic', r.get('traffic'))
print(json.dumps(d.get('kernel_families'))[:1500])
print(json.dumps(d.get('other_configs'), indent=0)[:2500])
print({k: d.get(k) for k in ('cpu_baseline',)})
"
timeout 500 python tools/bench_pipeline.py 64 3 2 4 8 > $O/pipeline.jsonl 2> $O/pipeline.err; cat $O/pipeline.jsonl; tail -3 $O/pipeline.err
timeout 900 python -m pytest tests/test_wrappers.py tests/test_boundary.py tests/test_parallel.py tests/test_engine.py -x -q -m gpu 2>&1 | tail -4 > $O/tests.log; cat $O/tests.log
