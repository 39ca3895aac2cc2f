# round 4, session p: one-launch GroupNorm passes of the deepest level reading only the replicas in use (4, not 32) with batched loads
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4p; mkdir -p $O
OLD=$PWD/pytorchdeeplearing_amd/lib/variants/libsegengine_oldnorm.so
run() { t=$1; shift
  env "$@" timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --launch stream > $O/$t.json 2> $O/$t.err
  python -c "
import json; d=json.loads(open('$O/$t.json').read().strip().splitlines()[-1]); print('$t', d['value'], d['ms_per_step'], 'gn_small', d['kernel_families'].get('gn_small'))"
}
{
run new_1 SEG_SUB_MB=0
run old_1 SEGENGINE_LIB=$OLD
run new_2 SEG_SUB_MB=0
run old_2 SEGENGINE_LIB=$OLD
} 2>&1 | tee $O/ab.log
timeout 300 python -m pytest tests/test_engine.py -x -q -m gpu -k "parity" 2>&1 | tail -2
