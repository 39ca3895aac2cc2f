# round 4, session e: kernel-trace timelines of one step with and without the weight-gradient stream (which main-queue kernels pay for the overlap?)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4e; mkdir -p $O
for arm in side noside; do
  rm -rf gpurun_out/trace_$arm
  if [ $arm = noside ]; then export SEG_WGRAD_STREAM=0; else unset SEG_WGRAD_STREAM; fi
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_$arm -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --roofline-steps 0 --condition-seconds 0.2 --launch stream > $O/run_$arm.log 2>&1
  CSV=$(find gpurun_out/trace_$arm -name "*kernel_trace.csv" | head -1)
  python - "$CSV" $O/kernels_$arm.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last ~2 steps worth of kernels (compact)
rows = rows[-700:]
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["queue", "start_ns", "end_ns", "grid", "wg", "name"])
for r in rows:
    w.writerow([r["Queue_Id"], r["Start_Timestamp"], r["End_Timestamp"], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", ""), r["Kernel_Name"][:110]])
PY
  python tools/trace_gaps.py $CSV > $O/timeline_$arm.txt 2>&1
  rm -rf gpurun_out/trace_$arm
done
head -8 $O/timeline_side.txt; head -8 $O/timeline_noside.txt
