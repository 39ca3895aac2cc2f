# round 2, GPU session F: full per-kernel timeline of one step
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/trace
timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --roofline-steps 0 > gpurun_out/trace_run.log 2>&1
T=$(find gpurun_out/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $T > gpurun_out/r2f_trace_timeline.txt 2>&1
rm -rf gpurun_out/trace
wc -l gpurun_out/r2f_trace_timeline.txt
