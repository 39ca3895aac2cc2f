# Profile set of BASELINE configs[3] (C4: UNet3d 2 x 128^3 f16, 4 classes) the way the headline config has one: rocprofv3 kernel stats, the two HBM PMC passes,
# an un-instrumented kernel-trace timeline and the bytes per window of the step.  Outputs: gpurun_out/r6s/c4_*.
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r6s; mkdir -p $O
CMD="python tools/bench_configs.py"; export SEG_BENCH_ONLY=C4 SEG_BENCH_NOPROF=1
python -c "from pytorchdeeplearing_amd import _capi; print(_capi.product_library().build_info())" > $O/build.txt 2>&1
rm -rf gpurun_out/pmc gpurun_out/prof gpurun_out/trace
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc -o $c -- $CMD > $O/pmc_$c.log 2>&1
done
python profiles/summarize_pmc.py gpurun_out/pmc $O/c4_pmc_fetch_write_per_kernel.json > $O/pmc_summary.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o step -- $CMD > $O/prof_run.log 2>&1
DB=$(find gpurun_out/prof -name "*.db" | head -1)
if [ -n "$DB" ]; then python profiles/summarize_rocpd.py $DB 60 > $O/c4_rocprofv3_kernel_stats.txt 2>&1; fi
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- $CMD > $O/trace_run.log 2>&1
CSV=$(find gpurun_out/trace -name "*kernel_trace.csv" | head -1)
if [ -n "$CSV" ]; then python tools/trace_gaps.py $CSV > $O/c4_trace_timeline.txt 2>&1; python tools/window_bw.py $CSV gpurun_out/pmc 5.06 > $O/c4_window_bw.txt 2>&1; fi
rm -rf gpurun_out/pmc gpurun_out/prof gpurun_out/trace
cat $O/build.txt; tail -4 $O/pmc_summary.log; head -25 $O/c4_rocprofv3_kernel_stats.txt | cut -c1-150; head -30 $O/c4_window_bw.txt
