# round 5, session t: per-queue timeline of the un-instrumented UNet3d 2 x 128^3 step (C4): which queue carries the tail
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r5t; mkdir -p $O
rm -rf gpurun_out/trace_c4
SEG_BENCH_ONLY=C4 SEG_BENCH_NOPROF=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_c4 -o t -- python tools/bench_configs.py > $O/trace_run.log 2>&1
CSV=$(find gpurun_out/trace_c4 -name "*kernel_trace.csv" | head -1)
if [ -n "$CSV" ]; then python tools/trace_gaps.py $CSV > $O/trace_timeline_c4.txt 2>&1; head -20 $O/trace_timeline_c4.txt; fi
rm -rf gpurun_out/trace_c4
