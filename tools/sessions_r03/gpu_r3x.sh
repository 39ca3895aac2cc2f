# round 3, session x: the driver's two launch lines on a fresh lease (plain and torch.distributed.run with one rank)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3x_$1; mkdir -p $O
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/plain.json 2> $O/plain.err; cut -c1-200 $O/plain.json
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/torchrun.json 2> $O/torchrun.err; tail -1 $O/torchrun.json | cut -c1-200
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -6 | tee $O/smi.txt
