# round 4, session r: four-slice stages in conv_igemm for the deepest (<= 128 workgroup) launches
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4r; mkdir -p $O
SEG_IGEMM_KS=4 timeout 200 python -m pytest tests/test_ops.py -x -q -m gpu -k "conv and not conv3 and not wgrad" 2>&1 | tail -2
run() { t=$1; shift
  env "$@" timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --launch stream > $O/$t.json 2> $O/$t.err
  python -c "
import json; d=json.loads(open('$O/$t.json').read().strip().splitlines()[-1]); print('$t', d['value'], d['ms_per_step'], 'generic_conv', d['kernel_families'].get('generic_conv',{}).get('ms_per_step'))"
}
{
run new_1 SEG_SUB_MB=0
run ks2_1 SEG_IGEMM_KS=2
run new_2 SEG_SUB_MB=0
run ks2_2 SEG_IGEMM_KS=2
} 2>&1 | tee $O/ab.log
timeout 200 python -m pytest tests/test_engine.py -x -q -m gpu -k "parity_f32_gpu or parity_lowp_gpu" 2>&1 | tail -2
