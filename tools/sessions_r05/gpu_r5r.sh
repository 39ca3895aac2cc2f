# round 5, session r: the bias-gradient column sum of the UNet up-convs moved to the weight-gradient queue (and at most 512 blocks of same-address atomics):
# UNet parity tests on the GPU, C4 / C2 / C3 / C5 step times (before: C4 4.38-4.44 ms in three calls of the previous binary), kernel statistics of C4
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r5r; mkdir -p $O
python -c "from pytorchdeeplearing_amd import _capi; print(_capi.product_library().build_info())" > $O/build.txt 2>&1; cat $O/build.txt
timeout 900 python -m pytest tests/test_engine.py tests/test_fullsize.py tests/test_wrappers.py -m gpu -x -q -k "unet or C4 or C1 or c1" > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2; do SEG_BENCH_ONLY=C2,C3,C4,C5 timeout 300 python tools/bench_configs.py 2>/dev/null | cut -c1-120 >> $O/configs.log; done; cat $O/configs.log
rm -rf gpurun_out/prof_C4
SEG_BENCH_ONLY=C4 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_C4 -o step -- python tools/bench_configs.py > $O/prof_C4.log 2>&1
DB=$(find gpurun_out/prof_C4 -name "*.db" | head -1)
if [ -n "$DB" ]; then python profiles/summarize_rocpd.py $DB 30 > $O/kernel_stats_C4.txt 2>&1; head -14 $O/kernel_stats_C4.txt; grep colsum $O/kernel_stats_C4.txt; fi
rm -rf gpurun_out/prof_C4
