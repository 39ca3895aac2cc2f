# round 5, session q: (a) the train step against the batch size at 96^3 (1, 2, 4 = the benchmark, 8, 16 volumes): the fixed, latency-bound part of the step;
# (b) rocprofv3 kernel statistics of the other BASELINE configs (C2 VNet2d 16x512^2, C4 UNet3d 2x128^3, C5 VNet3d 1x160^3 bf16)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r5q; mkdir -p $O
for b in 1 2 4 8 16; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --batch $b --no-cpu-baseline --no-other-configs --roofline-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'batch': $b, 'volumes_per_s': d['value'], 'ms_per_step': d['ms_per_step']}))" >> $O/batch_scaling.jsonl
done
cat $O/batch_scaling.jsonl
for c in C2 C4 C5; do
  rm -rf gpurun_out/prof_$c
  SEG_BENCH_ONLY=$c timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$c -o step -- python tools/bench_configs.py > $O/prof_$c.log 2>&1
  DB=$(find gpurun_out/prof_$c -name "*.db" | head -1)
  if [ -n "$DB" ]; then python profiles/summarize_rocpd.py $DB 30 > $O/kernel_stats_$c.txt 2>&1; head -12 $O/kernel_stats_$c.txt; fi
  rm -rf gpurun_out/prof_$c
done
