cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/ab.log
timeout 600 python -m pytest tests/test_engine.py tests/test_fullsize.py tests/test_boundary.py -m gpu -x -q 2>&1 | tail -4 >> gpurun_out/ab.log
run() { echo "== $*" >> gpurun_out/ab.log; env "$@" timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --roofline-steps 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*' >> gpurun_out/ab.log; }
run SEG_DUAL_GN=0
run SEG_DUAL_GN=1
run SEG_DUAL_GN=0
run SEG_DUAL_GN=1
cat gpurun_out/ab.log
