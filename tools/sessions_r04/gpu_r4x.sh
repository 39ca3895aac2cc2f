# round 4, session x: early release (default now) with the number carried by the data-gradient kernel that follows the release
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4x; mkdir -p $O
timeout 400 python -m pytest tests/test_engine.py -x -q -m gpu -k "flag_forks or bucketed or op_ranges or parity_lowp_gpu" 2>&1 | tail -3
run() { t=$1; shift
  env "$@" timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --roofline-steps 0 --launch stream > $O/$t.json 2> $O/$t.err
  python -c "
import json; d=json.loads(open('$O/$t.json').read().strip().splitlines()[-1]); print('$t', d['value'], d['ms_per_step'])"
}
{
run default_1 SEG_SUB_MB=0
run event_1 SEG_FORK=event
run late_1 SEG_FLUSH_LATE=1
run batch1 SEG_FORK_BATCH=1
run batch2 SEG_FORK_BATCH=2
run batch4 SEG_FORK_BATCH=4
run heavy4 SEG_FORK_HEAVY_MB=4
run heavy64 SEG_FORK_HEAVY_MB=64
run default_2 SEG_SUB_MB=0
run event_2 SEG_FORK=event
run tail1 SEG_TAIL_WGRADS=1
run batch1_heavy SEG_FORK_BATCH=1 SEG_FORK_HEAVY_MB=4
} 2>&1 | tee $O/ab.log
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_x -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --roofline-steps 0 --condition-seconds 0.2 --launch stream > $O/trace_run.log 2>&1
CSV=$(find gpurun_out/trace_x -name "*kernel_trace.csv" | head -1)
if [ -n "$CSV" ]; then python tools/trace_gaps.py $CSV > $O/trace_timeline.txt 2>&1; fi
rm -rf gpurun_out/trace_x; head -14 $O/trace_timeline.txt
