# round 5, session e: per-XCD weight-gradient tiles (SEG_W3_XCD=1, default) against per-workgroup partial tiles (=0): operator tests with the path forced,
# full-size gradient tests, in-call A/B of the driver's command, kernel stats and the two HBM PMC passes of the new path
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r5e; mkdir -p $O
SEG_W3_XCD=2 SEG_W3_MINBOX=1 timeout 600 python -m pytest tests/test_ops.py -m gpu -x -q -k "wgrad3 and not wgrad3x_kernel" > $O/ops_xcd.log 2>&1; tail -2 $O/ops_xcd.log
timeout 900 python -m pytest tests/test_fullsize.py -m gpu -x -q -k "gradients" > $O/fullsize_grad.log 2>&1; tail -3 $O/fullsize_grad.log
DRV="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs"
for arm in 0 1 0 1; do
  echo "== SEG_W3_XCD=$arm" >> $O/xcd_ab.log
  SEG_W3_XCD=$arm timeout 300 $DRV 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:(v['ms_per_step'],v['frac']) for k,v in d['kernel_families'].items() if 'wgrad' in k})" >> $O/xcd_ab.log
done
cat $O/xcd_ab.log
SHORT="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --roofline-steps 0 --condition-seconds 0 --launch stream"
rm -rf gpurun_out/pmc gpurun_out/prof
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc -o $c -- $SHORT > $O/pmc_$c.log 2>&1
done
python profiles/summarize_pmc.py gpurun_out/pmc $O/pmc_fetch_write_per_kernel.json > $O/pmc_summary.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o step -- $DRV --roofline-steps 0 --launch stream > $O/prof_run.log 2>&1
DB=$(find gpurun_out/prof -name "*.db" | head -1)
if [ -n "$DB" ]; then python profiles/summarize_rocpd.py $DB 60 > $O/rocprofv3_kernel_stats.txt 2>&1; fi
rm -rf gpurun_out/pmc gpurun_out/prof
grep -i "wgrad3\|step GPU" $O/rocprofv3_kernel_stats.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5e/pmc_fetch_write_per_kernel.json'))
for k,v in d.items():
    if 'wgrad3' in k: print(k[:60], v)
print(d['_total'])
PY
