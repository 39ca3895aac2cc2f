cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r6t; mkdir -p $O
python -c "from pytorchdeeplearing_amd import _capi; print(_capi.product_library().build_info())" > $O/build.txt 2>&1; cat $O/build.txt
timeout 600 python -m pytest tests/test_conv3x.py -m gpu -x -q > $O/test_conv3x.log 2>&1; tail -3 $O/test_conv3x.log
timeout 300 python tools/bench_conv3x_cfgs.py "4,96,16,16:25,29,24,28" "2,128,16,16:25,29,24,28" "2,128,16,32:26,27,30,31" "1,160,16,16:25,29" "2,64,16,16:24,28,25,29" > $O/standalone.jsonl 2> $O/standalone.err
cat $O/standalone.jsonl; tail -3 $O/standalone.err
AB="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --roofline-steps 0"
G='"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*'
for i in 1 2 3; do
for cfg in "SEG_C3X16_REUSE=0" "SEG_C3X16_REUSE=1"; do
  echo "== $cfg ($i)" >> $O/ab.log; env $cfg timeout 200 $AB 2>/dev/null | grep -o "$G" >> $O/ab.log
  env $cfg SEG_BENCH_ONLY=C4,C5 SEG_BENCH_NOPROF=1 timeout 200 python tools/bench_configs.py 2>/dev/null | cut -c1-90 >> $O/ab.log
done; done
cat $O/ab.log
