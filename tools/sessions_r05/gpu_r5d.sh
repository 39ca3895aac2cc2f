# round 5, session d: 8 x 8 x 32 tiles for the clDice kernels (SEG_SKEL_TZ=8) against 4 x 8 x 32, and the repaired new tests
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r5d; mkdir -p $O
for arm in "SEG_SKEL_TZ=4" "SEG_SKEL_TZ=8" "SEG_SKEL_TZ=4" "SEG_SKEL_TZ=8"; do
  echo "== $arm" >> $O/cldice_tz_ab.log
  env $arm timeout 200 python tools/prof_cldice_step.py 2>&1 | grep cldice_weight >> $O/cldice_tz_ab.log
done
cat $O/cldice_tz_ab.log
SEG_SKEL_TZ=8 timeout 600 python -m pytest tests/test_cldice.py -m gpu -x -q > $O/cldice_tests_tz8.log 2>&1; tail -2 $O/cldice_tests_tz8.log
SEG_FULLSIZE_REPORT=$O/fullsize_report.txt timeout 900 python -m pytest tests/test_fullsize.py -m gpu -q -k "C1 or c1_unet2d or three_hundred" > $O/new_tests.log 2>&1
tail -6 $O/new_tests.log; cat $O/fullsize_report.txt
