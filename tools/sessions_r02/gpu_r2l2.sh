cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/trace
SEG_C3X_TWICE=1 SEG_WGRAD_STREAM=0 timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --roofline-steps 0 > gpurun_out/trace_run.log 2>&1
T=$(find gpurun_out/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $T > gpurun_out/r2l_trace_twice.txt 2>&1
rm -rf gpurun_out/trace
tail -3 gpurun_out/r2l_trace_twice.txt | cut -c1-150
