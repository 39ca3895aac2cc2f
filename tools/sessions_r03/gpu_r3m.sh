# round 3, session m: confirmation of the final tree (GPU suite, smoke, the driver's bench command)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3m; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/gpu_tests.log; cat $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json
timeout 300 python tools/bench_configs.py 2>/dev/null | cut -c1-110
