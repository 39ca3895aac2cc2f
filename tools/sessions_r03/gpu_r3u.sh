# round 3, session u: what a fork point costs on the main queue (event record vs stream write/wait value), default policy confirmation
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3u; mkdir -p $O
timeout 120 tools/microbench/fork_cost 2>&1 | tee $O/fork_cost.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/base.json 2> $O/base.err; cut -c1-200 $O/base.json
