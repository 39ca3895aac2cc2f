# round 4, session a: first measurements of the round.  (1) new GPU tests (sub-batched chains, big-box weight gradient), (2) XCD-local barrier
# microbenchmark, (3) A/B inside one lease: baseline / sub-batched finest level (1, 2 samples per group) / 16-channel weight gradient with the
# old box / SEG_FORK_FLAG / wgrad3x at one workgroup per CU, (4) fork-flag stress, (5) standalone weight-gradient timings
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4a; mkdir -p $O
timeout 600 python -m pytest tests/test_engine.py tests/test_ops.py -x -q -m gpu -k "sub_batched or big_box or wgrad3" 2>&1 | tail -4 > $O/tests.log; cat $O/tests.log
timeout 120 tools/microbench/xcd_barrier 2000 > $O/xcd_barrier.log 2>&1; cat $O/xcd_barrier.log
run() { # tag, env...
  t=$1; shift
  env "$@" timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --launch stream > $O/$t.json 2> $O/$t.err
  python -c "
import json; d=json.loads(open('$O/$t.json').read().strip().splitlines()[-1]); print('$t', d['value'], d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'])"
}
for rep in 1 2; do
  run base_$rep SEG_SUB_MB=0
  run sub1_$rep SEG_SUB_MB=32
  run sub2_$rep SEG_SUB_MB=64
  run oldbox_$rep SEG_SUB_MB=0 SEG_W3_BOX16=0
  run sub1_oldbox_$rep SEG_SUB_MB=32 SEG_W3_BOX16=0
  run flag_$rep SEG_SUB_MB=0 SEG_FORK_FLAG=1
  run w3x_$rep SEG_SUB_MB=0 SEG_WGRAD3X=1 SEG_W3X_TOTAL=256
done 2>&1 | tee $O/ab.log
timeout 200 python tools/stress_fork_flag.py 10000 > $O/fork_flag_stress.json 2> $O/fork_flag_stress.err; tail -c 1500 $O/fork_flag_stress.json
timeout 200 python tools/bench_wgrad3.py SEG_W3_BOX16=0 SEG_W3_BOX16=1 SEG_WGRAD3X=1,SEG_W3X_TOTAL=256 > $O/wgrad3_standalone.log 2>&1; cat $O/wgrad3_standalone.log
