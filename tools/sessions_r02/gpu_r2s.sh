# round 2, GPU session S: bucketed exchange without main-stream stalls (loop-back, RCCL one-rank communicator, stall measurement)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_engine.py tests/test_parallel.py -m gpu -x -q -k "bucketed or op_ranges or rccl" 2>&1 | tail -6
timeout 300 python tools/bench_bucket_stall.py 2>&1 | grep variant | tee gpurun_out/r2s_bucket_stall.jsonl
