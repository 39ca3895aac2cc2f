# round 4, session b: the pipelined, branch-free wgrad3_kernel (32 x 16 tiles) against the round-3 kernel: bit-exact GPU tests, standalone timings per
# layer shape (+ a rocprofv3 kernel split of the new one), and the train step A/B inside one lease
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4b; mkdir -p $O
OLD=$PWD/pytorchdeeplearing_amd/lib/variants/libsegengine_oldw3.so
timeout 600 python -m pytest tests/test_ops.py tests/test_engine.py -x -q -m gpu -k "wgrad3 or big_box or parity_lowp_gpu" 2>&1 | tail -4 > $O/tests.log; cat $O/tests.log
SEG_W3_CQ=32 timeout 300 python -m pytest tests/test_ops.py -x -q -m gpu -k "wgrad3" 2>&1 | tail -2 > $O/tests_cq32.log; cat $O/tests_cq32.log
timeout 400 python tools/bench_wgrad3.py SEGENGINE_LIB=$OLD,SEG_W3_BOX16=0 SEG_W3_BOX16=0 SEG_W3_BOX16=1 SEG_W3_CQ=32,SEG_W3_BOX16=1 > $O/wgrad3_standalone.log 2>&1; cat $O/wgrad3_standalone.log
rm -rf gpurun_out/prof_w3
timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_w3 -o w3 -- python tools/bench_wgrad3.py child > $O/prof_run.log 2>&1
DB=$(find gpurun_out/prof_w3 -name "*.db" | head -1)
if [ -n "$DB" ]; then python profiles/summarize_rocpd.py $DB 30 > $O/wgrad3_kernel_split.txt 2>&1; head -20 $O/wgrad3_kernel_split.txt; fi
rm -rf gpurun_out/prof_w3
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
