# round 6, session a (first call, HEAD binary 763f887feae4 = round 5's final): the lease's baseline bench lines and the 16-bit gradient calibration against
# torch.autocast (measurement run: SEG_LOWP_CAL lifted so that all four cases print their three columns)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6a; mkdir -p $O
python -c "from pytorchdeeplearing_amd import _capi; print(_capi.product_library().build_info())" > $O/build.txt 2>&1; cat $O/build.txt
DRV="python bench.py --gpus 1 --steps 20 --warmup 5"
timeout 300 $DRV --no-cpu-baseline --no-other-configs > $O/bench_1.json 2> $O/bench_1.err
SEG_LOWP_CAL=1000 SEG_FULLSIZE_REPORT=$O/calibration.txt timeout 900 python -m pytest tests/test_fullsize.py -m gpu -q -k "calibrated" -s > $O/calibration.log 2>&1
timeout 300 $DRV --no-cpu-baseline --no-other-configs > $O/bench_2.json 2> $O/bench_2.err
tail -5 $O/calibration.log; cat $O/calibration.txt; for i in 1 2; do cut -c1-200 $O/bench_$i.json; done
