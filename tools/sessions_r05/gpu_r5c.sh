# round 5, session c: soft-clDice kernels - bit-image target skeleton and the four-voxels-per-thread tile kernels against the round-4 kernels (env switches,
# one process each), GPU parity tests of the clDice path, kernel stats of the C5 + clDice step with the new kernels, the new full-size tests
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r5c; mkdir -p $O
timeout 600 python -m pytest tests/test_cldice.py -m gpu -x -q > $O/cldice_tests.log 2>&1; tail -3 $O/cldice_tests.log
for arm in "SEG_SKEL_X4=0 SEG_CLD_BITS=0" "SEG_SKEL_X4=1 SEG_CLD_BITS=0" "SEG_SKEL_X4=0 SEG_CLD_BITS=1" "SEG_SKEL_X4=1 SEG_CLD_BITS=1" "SEG_SKEL_X4=0 SEG_CLD_BITS=0" "SEG_SKEL_X4=1 SEG_CLD_BITS=1"; do
  echo "== $arm" >> $O/cldice_ab.log
  env $arm timeout 200 python tools/prof_cldice_step.py >> $O/cldice_ab.log 2>&1
done
cat $O/cldice_ab.log
rm -rf gpurun_out/profc
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/profc -o step -- python tools/prof_cldice_step.py > $O/prof_cldice_run.log 2>&1
DB=$(find gpurun_out/profc -name "*.db" | head -1)
if [ -n "$DB" ]; then python profiles/summarize_rocpd.py $DB 12 > $O/kernel_stats_c5_cldice.txt 2>&1; fi
rm -rf gpurun_out/profc
head -14 $O/kernel_stats_c5_cldice.txt
SEG_FULLSIZE_REPORT=$O/fullsize_report.txt timeout 900 python -m pytest tests/test_fullsize.py -m gpu -q -k "C1 or c1_unet2d or three_hundred or cldice" --durations=8 > $O/new_tests.log 2>&1
tail -15 $O/new_tests.log; cat $O/fullsize_report.txt
