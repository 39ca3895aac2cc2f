# round 4, session s: flag forks (hipStreamWaitValue32 on the weight-gradient queue, the number stored by the main queue's next kernel) vs event forks
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4s; mkdir -p $O
timeout 300 python -m pytest tests/test_engine.py -x -q -m gpu -k "flag_forks or bucketed or op_ranges" 2>&1 | tail -3
run() { t=$1; shift
  env "$@" timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --launch stream > $O/$t.json 2> $O/$t.err
  python -c "
import json; d=json.loads(open('$O/$t.json').read().strip().splitlines()[-1]); print('$t', d['value'], d['ms_per_step'])"
}
{
run flag_1 SEG_SUB_MB=0
run event_1 SEG_FORK=event
run flag_2 SEG_SUB_MB=0
run event_2 SEG_FORK=event
} 2>&1 | tee $O/ab.log
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_s -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --roofline-steps 0 --condition-seconds 0.2 --launch stream > $O/trace_run.log 2>&1
CSV=$(find gpurun_out/trace_s -name "*kernel_trace.csv" | head -1)
if [ -n "$CSV" ]; then python tools/trace_gaps.py $CSV > $O/trace_timeline.txt 2>&1; fi
rm -rf gpurun_out/trace_s; head -24 $O/trace_timeline.txt
