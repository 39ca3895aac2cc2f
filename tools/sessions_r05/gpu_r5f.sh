# round 5, session f: after the product / experiments split - phase trace of conv3x_kernel (diagnostic variant), kernel-trace timeline of the HIP-graph replay of the
# step (why is it slower than stream launches?), the driver's command twice, the whole GPU parity suite against the product library
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r5f; mkdir -p $O
timeout 200 python tools/trace_conv3x.py 2> $O/conv3x_phase_trace.log > /dev/null; grep "conv3x trace" $O/conv3x_phase_trace.log | awk 'NR%3==0'
DRV="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs"
for i in 1 2; do timeout 300 $DRV > $O/bench_$i.json 2> $O/bench_$i.err; cut -c1-200 $O/bench_$i.json; done
rm -rf gpurun_out/trace_g
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_g -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --roofline-steps 0 --condition-seconds 0.2 --launch graph > $O/trace_graph_run.log 2>&1
CSV=$(find gpurun_out/trace_g -name "*kernel_trace.csv" | head -1)
if [ -n "$CSV" ]; then python tools/trace_gaps.py $CSV > $O/trace_timeline_graph_replay.txt 2>&1; fi
rm -rf gpurun_out/trace_g
head -8 $O/trace_timeline_graph_replay.txt
SEG_FULLSIZE_REPORT=$O/fullsize_report.txt timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > $O/gpu_tests_full.log 2>&1; tail -22 $O/gpu_tests_full.log
