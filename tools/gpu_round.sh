set -x
mkdir -p gpurun_out
cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/gpu_tests.log; echo "rc=$?" >> gpurun_out/gpu_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err
timeout 300 python tools/bench_prepost.py > gpurun_out/prepost.jsonl 2> gpurun_out/prepost.err
timeout 300 python tools/bench_cldice.py > gpurun_out/cldice.jsonl 2> gpurun_out/cldice.err
tail -3 gpurun_out/gpu_tests.log; cat gpurun_out/bench.json | cut -c1-400; cat gpurun_out/prepost.jsonl; cat gpurun_out/cldice.jsonl; tail -3 gpurun_out/*.err
