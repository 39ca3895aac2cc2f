# round 4, session u: hipStreamWaitValue32 as a COMMAND-PROCESSOR wait (rocclr flag GPU_STREAMOPS_CP_WAIT=1: barrier-value packet instead of the
# one-thread __amd_rocclr_streamOpsWait kernel that session s found on the weight-gradient queue)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4u; mkdir -p $O
GPU_STREAMOPS_CP_WAIT=1 SEG_FORK=flag timeout 300 python -m pytest tests/test_engine.py -x -q -m gpu -k "flag_forks" 2>&1 | tail -2
run() { t=$1; shift
  env "$@" timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --launch stream > $O/$t.json 2> $O/$t.err
  python -c "
import json; d=json.loads(open('$O/$t.json').read().strip().splitlines()[-1]); print('$t', d['value'], d['ms_per_step'])"
}
{
run cpwait_1 SEG_FORK=flag GPU_STREAMOPS_CP_WAIT=1
run event_1 SEG_FORK=event
run cpwait_2 SEG_FORK=flag GPU_STREAMOPS_CP_WAIT=1
run event_2 SEG_FORK=event
} 2>&1 | tee $O/ab.log
GPU_STREAMOPS_CP_WAIT=1 SEG_FORK=flag timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_u -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --roofline-steps 0 --condition-seconds 0.2 --launch stream > $O/trace_run.log 2>&1
CSV=$(find gpurun_out/trace_u -name "*kernel_trace.csv" | head -1)
if [ -n "$CSV" ]; then python tools/trace_gaps.py $CSV > $O/trace_timeline.txt 2>&1; fi
rm -rf gpurun_out/trace_u; head -12 $O/trace_timeline.txt
