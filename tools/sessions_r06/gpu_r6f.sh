# round 6, session f: halo-conv tilings 60 / 61 (two wave groups per workgroup, each walking half of the resident chunks) at the 12^3 / 6^3 levels: operator tests,
# standalone times against tilings 7 / 45, and the train step with SEG_C3X_MAP selecting them (one binary, alternating runs)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6f; mkdir -p $O
python -c "from pytorchdeeplearing_amd import _capi; print(_capi.product_library().build_info())" > $O/build.txt 2>&1; cat $O/build.txt
timeout 900 python -m pytest tests/test_conv3x.py -m gpu -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 300 python tools/bench_conv3x_cfgs.py "4,12,128,128:7,60,44" "4,6,256,256:45,61,7,60" "4,24,64,64:3,60" "1,20,128,128:7,60" "1,10,256,256:45,61" "2,16,128,128:7,60" "2,8,256,256:45,61" > $O/standalone.jsonl 2> $O/standalone.err; cat $O/standalone.jsonl
AB="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --roofline-steps 0"
G='"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*'
for i in 1 2 3; do
  echo "== default ($i)" >> $O/ab.log; timeout 200 $AB 2>/dev/null | grep -o "$G" >> $O/ab.log
  echo "== 12^3 -> 60, 6^3 -> 61 ($i)" >> $O/ab.log; SEG_C3X_MAP="128:128:12=60,256:256:6=61" timeout 200 $AB 2>/dev/null | grep -o "$G" >> $O/ab.log
  echo "== 12^3 -> 60 ($i)" >> $O/ab.log; SEG_C3X_MAP="128:128:12=60" timeout 200 $AB 2>/dev/null | grep -o "$G" >> $O/ab.log
  echo "== 6^3 -> 61 ($i)" >> $O/ab.log; SEG_C3X_MAP="256:256:6=61" timeout 200 $AB 2>/dev/null | grep -o "$G" >> $O/ab.log
done
cat $O/ab.log
