cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r6q; mkdir -p $O
rm -rf gpurun_out/trace
SEG_BENCH_ONLY=C4 SEG_BENCH_NOPROF=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python tools/bench_configs.py > $O/trace_run.log 2>&1
CSV=$(find gpurun_out/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $CSV > $O/c4_trace_timeline.txt 2>&1
rm -rf gpurun_out/trace
tail -3 $O/trace_run.log; head -60 $O/c4_trace_timeline.txt
