# round 5, session n: the release schedule of the weight-gradient queue re-tuned with this round's kernels (the batch size was last swept in round 2).  Experiments
# library (the knobs are compiled out of the product), one call, baseline interleaved: SEG_FORK_BATCH, SEG_FORK_HEAVY_MB, SEG_TAIL_WGRADS, SEG_FLUSH_LATE, SEG_SIDE_PRIO
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r5n; mkdir -p $O
export SEGENGINE_LIB=$PWD/pytorchdeeplearing_amd/lib/libsegengine_exp.so
DRV="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --roofline-steps 0"
run() { echo -n "$1: " >> $O/sweep.log; env $1 timeout 300 $DRV 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])' >> $O/sweep.log; }
for arm in "X=0" "SEG_FORK_BATCH=2" "SEG_FORK_BATCH=4" "SEG_FORK_BATCH=6" "X=0" "SEG_FORK_BATCH=9" "SEG_FORK_HEAVY_MB=4" "SEG_FORK_HEAVY_MB=8" "SEG_FORK_HEAVY_MB=32" "X=0" "SEG_FORK_HEAVY_MB=100000" "SEG_TAIL_WGRADS=1" "SEG_FLUSH_LATE=1" "SEG_SIDE_PRIO=0" "X=0" "SEG_FORK_BATCH=4 SEG_FORK_HEAVY_MB=8" "SEG_FORK_BATCH=2 SEG_FORK_HEAVY_MB=32"; do run "$arm"; done
cat $O/sweep.log
