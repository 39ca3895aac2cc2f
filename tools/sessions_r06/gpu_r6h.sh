# round 6, session h: (1) both data-gradients of the 1^d conv on the concat from one pass over d(raw) (seg_conv_args.out1), (2) the 1^d head inside the activation pass
# that writes its input (SEG_HEAD_FUSE): tests, the bench against the previous binary (f7c335a) and with SEG_HEAD_FUSE=0, the whole GPU parity suite, other configs
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6h; mkdir -p $O
python -c "from pytorchdeeplearing_amd import _capi; print(_capi.product_library().build_info())" > $O/build.txt 2>&1; cat $O/build.txt
timeout 900 python -m pytest tests/test_ops.py tests/test_engine.py -m gpu -q -x -k "two_outputs or head_inside or activation or abi or parity_f32_gpu or virtual_head" > $O/tests.log 2>&1; tail -3 $O/tests.log
AB="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --roofline-steps 0"
G='"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*'
PREV=tools/experiments/libsegengine_prev.so
for i in 1 2 3; do
  echo "== new ($i)" >> $O/ab.log; timeout 200 $AB 2>/dev/null | grep -o "$G" >> $O/ab.log
  echo "== new, SEG_HEAD_FUSE=0 ($i)" >> $O/ab.log; SEG_HEAD_FUSE=0 timeout 200 $AB 2>/dev/null | grep -o "$G" >> $O/ab.log
  echo "== prev ($i)" >> $O/ab.log; SEGENGINE_LIB=$PREV timeout 200 $AB 2>/dev/null | grep -o "$G" >> $O/ab.log
done
cat $O/ab.log
SEG_BENCH_ONLY=C4,C5,C2 timeout 300 python tools/bench_configs.py > $O/configs_new.jsonl 2> $O/configs_new.err
SEGENGINE_LIB=$PREV SEG_BENCH_ONLY=C4,C5,C2 timeout 300 python tools/bench_configs.py > $O/configs_prev.jsonl 2> $O/configs_prev.err
cut -c1-120 $O/configs_new.jsonl $O/configs_prev.jsonl
SEG_FULLSIZE_REPORT=$O/fullsize_report.txt timeout 1800 python -m pytest tests -m gpu -q --durations=10 > $O/gpu_tests_full.log 2>&1; tail -15 $O/gpu_tests_full.log
