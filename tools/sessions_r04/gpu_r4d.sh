# round 4, session d: wgrad3_kernel with the prefetch loads interleaved into the (fully unrolled, pipelined) sweep: phase trace, standalone, step A/B
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4d; mkdir -p $O
OLD=$PWD/pytorchdeeplearing_amd/lib/variants/libsegengine_oldw3.so
for arm in "SEG_W3_BOX16=0" "SEG_W3_BOX16=1" "SEG_W3_CQ=32"; do
  echo "== $arm"; env $arm timeout 120 python tools/trace_wgrad3.py 2>&1 | grep "wgrad3 trace" | awk 'NR%3==0'
done > $O/trace.log 2>&1
cat $O/trace.log
timeout 300 python -m pytest tests/test_ops.py -x -q -m gpu -k "wgrad3 or big_box" 2>&1 | tail -2 > $O/tests.log; cat $O/tests.log
timeout 400 python tools/bench_wgrad3.py SEGENGINE_LIB=$OLD,SEG_W3_BOX16=0 SEG_W3_BOX16=0 SEG_W3_BOX16=1 SEG_W3_CQ=32,SEG_W3_BOX16=1 > $O/wgrad3_standalone.log 2>&1; cat $O/wgrad3_standalone.log
run() { # tag, env...
  t=$1; shift
  env "$@" timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --launch stream > $O/$t.json 2> $O/$t.err
  python -c "
import json; d=json.loads(open('$O/$t.json').read().strip().splitlines()[-1]); print('$t', d['value'], d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'])"
}
for rep in 1 2; do
  run old_$rep SEGENGINE_LIB=$OLD SEG_W3_BOX16=0
  run new_$rep SEG_W3_BOX16=1
  run new_smallbox_$rep SEG_W3_BOX16=0
  run new_cq32_$rep SEG_W3_CQ=32
done 2>&1 | tee $O/ab.log
