# the driver's multi-GPU launch line with ONE rank on the one GPU of the box: torch.distributed.run + RCCL initialisation + the bucketed exchange path of bench.py on a real (one-rank) communicator
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r6z; mkdir -p $O
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/torchrun_n1.json 2> $O/torchrun_n1.err
echo rc $?; cut -c1-400 $O/torchrun_n1.json; tail -3 $O/torchrun_n1.err
