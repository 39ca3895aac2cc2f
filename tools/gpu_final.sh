# Round-end measurement set on one MI355X: GPU parity tests, the contract bench line, rocprofv3 kernel stats of the same
# command, the two PMC passes (separate runs, kernel-trace only), the other BASELINE configs.  Outputs under gpurun_out/.
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-step24}
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/gpu_tests_$TAG.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; tail -3 gpurun_out/smoke_$TAG.log
timeout 300 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
rm -rf gpurun_out/prof gpurun_out/pmc
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o step -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --roofline-steps 0 > gpurun_out/prof_run.log 2>&1
DB=$(find gpurun_out/prof -name "*.db" | head -1)
if [ -n "$DB" ]; then python profiles/summarize_rocpd.py $DB 45 > gpurun_out/kernel_stats_$TAG.txt 2>&1; fi
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc -o $c -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --roofline-steps 0 > gpurun_out/pmc_$c.log 2>&1
done
python profiles/summarize_pmc.py gpurun_out/pmc gpurun_out/pmc_fetch_write_$TAG.json > gpurun_out/pmc_summary.log 2>&1
timeout 300 python tools/bench_configs.py > gpurun_out/configs_$TAG.jsonl 2> gpurun_out/configs.err
timeout 300 python tools/bench_prepost.py > gpurun_out/prepost_$TAG.jsonl 2> gpurun_out/prepost.err
find gpurun_out/pmc -name "*.csv" -size +5M -delete; rm -rf gpurun_out/prof
cat gpurun_out/gpu_tests_$TAG.log; cut -c1-300 gpurun_out/bench_$TAG.json; head -12 gpurun_out/kernel_stats_$TAG.txt; tail -3 gpurun_out/pmc_summary.log; cat gpurun_out/configs_$TAG.jsonl | cut -c1-120
