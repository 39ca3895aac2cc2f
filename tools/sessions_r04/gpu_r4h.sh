# round 4, session h: (1) fewer weight-gradient workgroups now that the side queue has slack, (2) the clDice config: bench_configs vs other_configs,
# (3) pipeline GPU test
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4h; mkdir -p $O
run() { t=$1; shift
  env "$@" timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --launch stream > $O/$t.json 2> $O/$t.err
  python -c "
import json; d=json.loads(open('$O/$t.json').read().strip().splitlines()[-1]); print('$t', d['value'], d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'])"
}
{
run base_1 SEG_SUB_MB=0
run w3_128 SEG_W3_TOTAL=128
run w3_192 SEG_W3_TOTAL=192
run w316_128 SEG_W3_TOTAL16=128
run w316_192 SEG_W3_TOTAL16=192
run wg_1024 SEG_WG_TOTAL=1024
run wg_512 SEG_WG_TOTAL=512
run all_192 SEG_W3_TOTAL=192 SEG_W3_TOTAL16=192
run base_2 SEG_SUB_MB=0
} 2>&1 | tee $O/ab.log
timeout 300 python tools/bench_configs.py > $O/configs.jsonl 2> $O/configs.err; cut -c1-200 $O/configs.jsonl
timeout 300 python -m pytest tests/test_wrappers.py -x -q -m gpu -k "pipeline_feeds" 2>&1 | tail -3
