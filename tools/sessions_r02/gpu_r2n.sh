# round 2, GPU session N: soft-clDice as an engine call
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_cldice.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python tools/bench_configs.py 2>/dev/null | grep "C5" | cut -c1-260 | tee gpurun_out/r2n_cldice.jsonl
