# round 2, GPU session L: cold vs warm operands of the halo convs inside the step (every conv launched twice); knob A/B
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/trace
SEG_C3X_TWICE=1 SEG_WGRAD_STREAM=0 timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --roofline-steps 0 > gpurun_out/trace_run.log 2>&1
T=$(find gpurun_out/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $T > gpurun_out/r2l_trace_twice.txt 2>&1
rm -rf gpurun_out/trace
awk '/full timeline/{f=1;next} f' gpurun_out/r2l_trace_twice.txt | grep "^q1" | grep -B1 -A0 "conv3x_kernel" | grep "conv3x_kernel" | awk '{print $3, $6}' > gpurun_out/r2l_pairs.txt
python - <<'PY'
import collections
rows=[l.split() for l in open('gpurun_out/r2l_pairs.txt')]
agg=collections.defaultdict(lambda:[0,0.0,0.0])
for i in range(0,len(rows)-1,2):
    (d1,g1),(d2,g2)=rows[i],rows[i+1]
    if g1!=g2: continue
    a=agg[g1]; a[0]+=1; a[1]+=float(d1); a[2]+=float(d2)
for g,(n,c,w) in sorted(agg.items()): print("%-18s pairs %2d  cold %.1f us  warm %.1f us"%(g,n,c/n,w/n))
PY
rm -f gpurun_out/r2l_ab.log
for cfg in "SEG_X=0" "SEG_FOLD_WGS=512" "SEG_FOLD_WGS=1024" "SEG_WGRAD3X=1" "SEG_FORK_HEAVY_MB=4"; do
  echo "== $cfg" >> gpurun_out/r2l_ab.log
  env $cfg timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --roofline-steps 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*' >> gpurun_out/r2l_ab.log
done
cat gpurun_out/r2l_ab.log
