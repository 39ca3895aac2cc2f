# Round-end measurement set on one MI355X (everything from the FINAL binary, the bench lines with the DRIVER'S command), measurements first, parity tests last under
# their own limit: smoke, the two HBM PMC passes (-> profiles/<prefix>_pmc_fetch_write_per_kernel.json, which bench.py reads for `traffic` if its build stamp matches
# the loaded library), two SQ counter passes (MFMA busy / MOPS / waves; LDS waits / conflicts), three fresh-process bench lines, rocprofv3 kernel stats, an
# un-instrumented kernel-trace timeline, bytes per window of the step, the copy-rate / contention microbenchmark, the other BASELINE configs, pre/post-processing,
# kernel stats of the f32 run dtype, then `pytest -m gpu`.  Outputs under gpurun_out/<tag>/; $2 (optional) = profiles/ prefix of the round, e.g. r05.
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-final}; O=gpurun_out/$TAG; mkdir -p $O
python -c "from pytorchdeeplearing_amd import _capi; print(_capi.product_library().build_info())" > $O/build.txt 2>&1; cat $O/build.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
DRV="python bench.py --gpus 1 --steps 20 --warmup 5"
SHORT="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --roofline-steps 0 --condition-seconds 0 --launch stream"
rm -rf gpurun_out/pmc gpurun_out/pmc_sq gpurun_out/prof gpurun_out/trace gpurun_out/prof32
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc -o $c -- $SHORT > $O/pmc_$c.log 2>&1
done
python profiles/summarize_pmc.py gpurun_out/pmc $O/pmc_fetch_write_per_kernel.json > $O/pmc_summary.log 2>&1
if [ -n "$2" ] && [ -s $O/pmc_fetch_write_per_kernel.json ]; then cp $O/pmc_fetch_write_per_kernel.json profiles/$2_pmc_fetch_write_per_kernel.json; fi
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_sq -o mfma -- $SHORT > $O/pmc_mfma.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_sq -o lds -- $SHORT > $O/pmc_lds.log 2>&1
python profiles/summarize_pmc_sq.py gpurun_out/pmc_sq $O/mfma_util_per_kernel.json mfma lds > $O/pmc_sq_summary.log 2>&1
for i in 1 2 3; do
  timeout 500 $DRV $( [ $i -gt 1 ] && echo --no-cpu-baseline --no-other-configs ) > $O/bench_driver_cmd_$i.json 2> $O/bench_driver_cmd_$i.err
done
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o step -- $DRV --no-cpu-baseline --no-other-configs --roofline-steps 0 --launch stream > $O/prof_run.log 2>&1
DB=$(find gpurun_out/prof -name "*.db" | head -1)
if [ -n "$DB" ]; then python profiles/summarize_rocpd.py $DB 60 > $O/rocprofv3_kernel_stats.txt 2>&1; fi
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --roofline-steps 0 --condition-seconds 0.2 --launch stream > $O/trace_run.log 2>&1
CSV=$(find gpurun_out/trace -name "*kernel_trace.csv" | head -1)
timeout 200 python tools/bench_contention.py > $O/contention.json 2> $O/contention.err
COPY=$(python -c "import json;print(json.load(open('$O/contention.json'))['a_copy_alone']['TBps_read_plus_write'])" 2>/dev/null)
if [ -n "$CSV" ]; then python tools/trace_gaps.py $CSV > $O/trace_timeline.txt 2>&1; python tools/window_bw.py $CSV gpurun_out/pmc $COPY > $O/window_bw.txt 2>&1; fi
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof32 -o step -- python bench.py --dtype f32 --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --roofline-steps 0 --condition-seconds 0.2 --launch stream > $O/prof32_run.log 2>&1
DB=$(find gpurun_out/prof32 -name "*.db" | head -1)
if [ -n "$DB" ]; then python profiles/summarize_rocpd.py $DB 40 > $O/rocprofv3_kernel_stats_f32.txt 2>&1; fi
timeout 300 python tools/bench_configs.py > $O/configs.jsonl 2> $O/configs.err
timeout 300 python tools/bench_prepost.py > $O/prepost.jsonl 2> $O/prepost.err
rm -rf gpurun_out/pmc gpurun_out/pmc_sq gpurun_out/prof gpurun_out/trace gpurun_out/prof32
# the parity tests LAST and under their own limit (round 4 lost its measurement set to a test run that ate the budget)
SEG_FULLSIZE_REPORT=$O/fullsize_report.txt timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $O/gpu_tests_full.log 2>&1; tail -25 $O/gpu_tests_full.log > $O/gpu_tests.log
cat $O/gpu_tests.log; for i in 1 2 3; do cut -c1-260 $O/bench_driver_cmd_$i.json; done; head -14 $O/rocprofv3_kernel_stats.txt; tail -3 $O/pmc_summary.log; head -8 $O/pmc_sq_summary.log; cat $O/contention.json; head -8 $O/window_bw.txt; cut -c1-140 $O/configs.jsonl; head -6 $O/trace_timeline.txt
