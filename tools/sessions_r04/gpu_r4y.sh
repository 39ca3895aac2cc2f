# round 4, session y: multi-step streaming kernel for the 1^d-conv weight gradients (wgrad_direct_kernel) vs wgrad_kernel (SEG_WG_DIRECT=0)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4y; mkdir -p $O
timeout 400 python -m pytest tests/test_ops.py -x -q -m gpu -k "test_wgrad_exact or test_wgrad_conv_transpose" 2>&1 | tail -3
run() { t=$1; shift
  env "$@" timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --roofline-steps 0 --launch stream > $O/$t.json 2> $O/$t.err
  python -c "
import json; d=json.loads(open('$O/$t.json').read().strip().splitlines()[-1]); print('$t', d['value'], d['ms_per_step'])"
}
{
run direct_1 SEG_SUB_MB=0
run old_1 SEG_WG_DIRECT=0
run flag_1 SEG_FORK=flag GPU_STREAMOPS_CP_WAIT=1
run direct_2 SEG_SUB_MB=0
run old_2 SEG_WG_DIRECT=0
run flag_2 SEG_FORK=flag GPU_STREAMOPS_CP_WAIT=1
} 2>&1 | tee $O/ab.log
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_y -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --roofline-steps 0 --condition-seconds 0.2 --launch stream > $O/trace_run.log 2>&1
CSV=$(find gpurun_out/trace_y -name "*kernel_trace.csv" | head -1)
if [ -n "$CSV" ]; then python tools/trace_gaps.py $CSV > $O/trace_timeline.txt 2>&1; fi
rm -rf gpurun_out/trace_y; head -8 $O/trace_timeline.txt; grep "wgrad_direct\|wgrad_kernel" $O/trace_timeline.txt | head -12
