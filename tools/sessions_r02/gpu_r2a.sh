# round 2, GPU session A: conv3x correctness on the GPU (bit-exact op tests), per-layer tiling table, step-level A/B
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv3x.py tests/test_ops.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r2a_ops_tests.log
cat gpurun_out/r2a_ops_tests.log
timeout 900 python tools/tune_conv3x.py --iters 20 > gpurun_out/r2a_tune_conv3x.jsonl 2> gpurun_out/r2a_tune.err
tail -1 gpurun_out/r2a_tune_conv3x.jsonl | cut -c1-1500; tail -3 gpurun_out/r2a_tune.err
for cfg in "SEG_CONV3X=0" "SEG_CONV3X=1"; do
  echo "== $cfg" >> gpurun_out/r2a_ab.log
  env $cfg timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --roofline-steps 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*' >> gpurun_out/r2a_ab.log
done
cat gpurun_out/r2a_ab.log
timeout 900 python -m pytest tests/test_engine.py tests/test_fullsize.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -60 > gpurun_out/r2a_engine_fullsize_tests.log
tail -45 gpurun_out/r2a_engine_fullsize_tests.log
