# round 5, session u: multi-class loss_backward with the per-class loops unrolled over the instantiation's slots (arrays in registers instead of scratch):
# loss / engine parity tests on the GPU, C2 / C4 step times (before: C4 4.28-4.31, C2 6.30-6.36 ms), the kernel's time in C4
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r5u; mkdir -p $O
python -c "from pytorchdeeplearing_amd import _capi; print(_capi.product_library().build_info())" > $O/build.txt 2>&1; cat $O/build.txt
timeout 900 python -m pytest tests/test_boundary.py tests/test_engine.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2 3; do SEG_BENCH_ONLY=C2,C4 timeout 300 python tools/bench_configs.py 2>/dev/null | cut -c1-120 >> $O/configs.log; done; cat $O/configs.log
rm -rf gpurun_out/prof_C4
SEG_BENCH_ONLY=C4 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_C4 -o step -- python tools/bench_configs.py > $O/prof_C4.log 2>&1
DB=$(find gpurun_out/prof_C4 -name "*.db" | head -1)
if [ -n "$DB" ]; then python profiles/summarize_rocpd.py $DB 40 > $O/kernel_stats_C4.txt 2>&1; grep "loss\|colsum\|head" $O/kernel_stats_C4.txt; fi
rm -rf gpurun_out/prof_C4
