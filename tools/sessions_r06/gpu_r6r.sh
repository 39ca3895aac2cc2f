cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r6r; mkdir -p $O
export SEGENGINE_LIB=pytorchdeeplearing_amd/lib/variants/libsegengine_b444.so
timeout 900 python tools/tune_conv3x.py --sets c3,c4,c5 --iters 30 > $O/tune.jsonl 2> $O/tune.err
python - <<'PY' > gpurun_out/r6r/summary.txt
import json, collections
rows=[json.loads(l) for l in open('gpurun_out/r6r/tune.jsonl') if l.startswith('{')]
by=collections.defaultdict(list)
for r in rows:
    if 'shape' in r and 'us' in r and r.get('cfg')!='conv3_kernel' and (r.get('same_as_conv3') or r.get('maxdiff',1)<1e-2): by[r['shape']].append(r)
m=[]
for k,v in by.items():
    d=[r for r in v if r['default']]; b=min(v,key=lambda r:r['us'])
    print(k, 'default', d[0]['cfg'] if d else None, d[0]['us'] if d else None, 'best', b['cfg'], b['us'], 'top3', [(r['cfg'],r['us']) for r in sorted(v,key=lambda r:r['us'])[:4]])
print(rows[-1].get('SEG_C3X_MAP'))
PY
cat $O/summary.txt
MAP=$(tail -1 $O/summary.txt)
AB="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --roofline-steps 0"
G='"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*'
for i in 1 2 3; do
for cfg in "SEG_NOP=1" "SEG_C3X_MAP=$MAP"; do
  echo "== $cfg ($i)" >> $O/ab.log; env $cfg timeout 200 $AB 2>/dev/null | grep -o "$G" >> $O/ab.log
  env $cfg SEG_BENCH_ONLY=C4,C5 SEG_BENCH_NOPROF=1 timeout 200 python tools/bench_configs.py 2>/dev/null | cut -c1-90 >> $O/ab.log
done; done
cat $O/ab.log
