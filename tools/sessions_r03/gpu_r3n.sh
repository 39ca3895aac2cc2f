# round 3, session n: separable min / max pools in the skeleton iteration (soft-clDice)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3n; mkdir -p $O
timeout 600 python -m pytest tests/test_cldice.py tests/test_fullsize.py -m gpu -x -q -k "cldice" 2>&1 | tail -3
timeout 300 python tools/bench_cldice.py 2>/dev/null | tail -6
timeout 300 python tools/bench_configs.py 2>/dev/null | grep -i cldice | cut -c1-200 | tee $O/configs_cldice.jsonl
