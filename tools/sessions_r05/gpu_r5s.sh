# round 5, session s: the bias-gradient column sum back on the main queue with the new kernel (<= 512 blocks of same-address atomics, four rows in flight):
# C4 step time (previous binary 4.38-4.44 ms; column sum on the weight-gradient queue 4.46-4.49), C1 UNet2d for the 2-D twin
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r5s; mkdir -p $O
python -c "from pytorchdeeplearing_amd import _capi; print(_capi.product_library().build_info())" > $O/build.txt 2>&1; cat $O/build.txt
for i in 1 2 3; do SEG_BENCH_ONLY=C4 timeout 300 python tools/bench_configs.py 2>/dev/null | cut -c1-330 >> $O/configs.log; done; cat $O/configs.log
