# round 5, session b: where the soft-clDice term's 2.06 ms go (kernel stats of the C5 + clDice step), and the three new GPU parity tests (C1 at its own size,
# 300-step 16-bit loss curves) with their measured numbers
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r5b; mkdir -p $O
rm -rf gpurun_out/profc
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/profc -o step -- python tools/prof_cldice_step.py > $O/prof_cldice_run.log 2>&1
DB=$(find gpurun_out/profc -name "*.db" | head -1)
if [ -n "$DB" ]; then python profiles/summarize_rocpd.py $DB 40 > $O/kernel_stats_c5_cldice.txt 2>&1; fi
rm -rf gpurun_out/profc
SEG_FULLSIZE_REPORT=$O/fullsize_report.txt timeout 900 python -m pytest tests/test_fullsize.py -m gpu -x -q -k "C1 or c1_unet2d or three_hundred" --durations=8 > $O/new_tests.log 2>&1
tail -4 $O/prof_cldice_run.log; head -30 $O/kernel_stats_c5_cldice.txt; tail -15 $O/new_tests.log; cat $O/fullsize_report.txt
