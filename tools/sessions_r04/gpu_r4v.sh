# round 4, session v: release policy under flag forks (a fork no longer idles the main queue: earlier / smaller releases?)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4v; mkdir -p $O
run() { t=$1; shift
  env "$@" timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --roofline-steps 0 --launch stream > $O/$t.json 2> $O/$t.err
  python -c "
import json; d=json.loads(open('$O/$t.json').read().strip().splitlines()[-1]); print('$t', d['value'], d['ms_per_step'])"
}
{
run default_1 SEG_SUB_MB=0
run early_1 SEG_FLUSH_LATE=0
run batch1 SEG_FORK_BATCH=1
run early_batch1 SEG_FLUSH_LATE=0 SEG_FORK_BATCH=1
run batch2 SEG_FORK_BATCH=2
run heavy4 SEG_FORK_HEAVY_MB=4
run tail1 SEG_TAIL_WGRADS=1
run default_2 SEG_SUB_MB=0
run early_2 SEG_FLUSH_LATE=0
run batch6 SEG_FORK_BATCH=6
} 2>&1 | tee $O/ab.log
