cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r6m; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -3
cut -c1-400 $O/bench_default.json
python - <<'P'
import json
d=json.load(open('gpurun_out/r6m/bench_default.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['n_gpus'], d['steps'], d['warmup'])
P
