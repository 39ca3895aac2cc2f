# Round-end measurement set on one MI355X (everything from the FINAL binary, the bench lines with the DRIVER'S command):
# smoke, three fresh-process bench lines, rocprofv3 kernel stats + kernel trace timeline of the same command, the two PMC passes (separate
# runs, kernel-trace only), the other BASELINE configs, pre/post-processing, full-size fidelity report.  Outputs under gpurun_out/<tag>/.
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-final}; O=gpurun_out/$TAG; mkdir -p $O       # $2 (optional): profiles/ prefix of the round, e.g. r04
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
DRV="python bench.py --gpus 1 --steps 20 --warmup 5"
# the PMC passes first: bench.py reads the families' HBM traffic from the newest profiles/rNN_pmc_fetch_write_per_kernel.json, which should be THIS binary's
rm -rf gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc -o $c -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --roofline-steps 0 --condition-seconds 0 --launch stream > $O/pmc_$c.log 2>&1
done
python profiles/summarize_pmc.py gpurun_out/pmc $O/pmc_fetch_write_per_kernel.json > $O/pmc_summary.log 2>&1
if [ -n "$2" ] && [ -s $O/pmc_fetch_write_per_kernel.json ]; then cp $O/pmc_fetch_write_per_kernel.json profiles/$2_pmc_fetch_write_per_kernel.json; fi
for i in 1 2 3; do
  timeout 400 $DRV $( [ $i -gt 1 ] && echo --no-cpu-baseline --no-other-configs ) > $O/bench_driver_cmd_$i.json 2> $O/bench_driver_cmd_$i.err
done
rm -rf gpurun_out/prof gpurun_out/pmc gpurun_out/trace
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o step -- $DRV --no-cpu-baseline --no-other-configs --roofline-steps 0 --launch stream > $O/prof_run.log 2>&1
DB=$(find gpurun_out/prof -name "*.db" | head -1)
if [ -n "$DB" ]; then python profiles/summarize_rocpd.py $DB 50 > $O/rocprofv3_kernel_stats.txt 2>&1; fi
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --roofline-steps 0 --condition-seconds 0.2 --launch stream > $O/trace_run.log 2>&1
CSV=$(find gpurun_out/trace -name "*kernel_trace.csv" | head -1)
if [ -n "$CSV" ]; then python tools/trace_gaps.py $CSV > $O/trace_timeline.txt 2>&1; fi
timeout 300 python tools/bench_configs.py > $O/configs.jsonl 2> $O/configs.err
timeout 300 python tools/bench_prepost.py > $O/prepost.jsonl 2> $O/prepost.err
# the parity tests LAST and under their own limit: in round 4 two new tests ran 48^3 cases on the host checker here and the whole call was cut off by the GPU budget
# before a single measurement had been taken (profiles/README.md)
SEG_FULLSIZE_REPORT=$O/fullsize_report.txt timeout 900 python -m pytest tests -m gpu -x -q --durations=15 > $O/gpu_tests_full.log 2>&1; tail -25 $O/gpu_tests_full.log > $O/gpu_tests.log
rm -rf gpurun_out/prof gpurun_out/pmc gpurun_out/trace
cat $O/gpu_tests.log; for i in 1 2 3; do cut -c1-260 $O/bench_driver_cmd_$i.json; done; head -14 $O/rocprofv3_kernel_stats.txt; tail -3 $O/pmc_summary.log; cut -c1-140 $O/configs.jsonl; head -12 $O/trace_timeline.txt
