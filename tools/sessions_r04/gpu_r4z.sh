# round 4, session z: step-boundary bookkeeping folded into neighbouring kernels (SEG_STEP_RIDERS=0: separate launches), unsplit weight re-pack
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4z; mkdir -p $O
timeout 500 python -m pytest tests/test_engine.py -x -q -m gpu -k "step_riders or train_steps or random_masks or graph_replay or bucketed" 2>&1 | tail -3
run() { t=$1; shift
  env "$@" timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --roofline-steps 0 --launch stream > $O/$t.json 2> $O/$t.err
  python -c "
import json; d=json.loads(open('$O/$t.json').read().strip().splitlines()[-1]); print('$t', d['value'], d['ms_per_step'])"
}
{
run riders_1 SEG_SUB_MB=0
run separate_1 SEG_STEP_RIDERS=0
run unsplit_1 SEG_PACK_SPLIT=0
run riders_2 SEG_SUB_MB=0
run separate_2 SEG_STEP_RIDERS=0
run unsplit_2 SEG_PACK_SPLIT=0
} 2>&1 | tee $O/ab.log
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_z -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --roofline-steps 0 --condition-seconds 0.2 --launch stream > $O/trace_run.log 2>&1
CSV=$(find gpurun_out/trace_z -name "*kernel_trace.csv" | head -1)
if [ -n "$CSV" ]; then python tools/trace_gaps.py $CSV > $O/trace_timeline.txt 2>&1; fi
rm -rf gpurun_out/trace_z; head -4 $O/trace_timeline.txt; grep "^q1" $O/trace_timeline.txt | head -12; grep "^q1" $O/trace_timeline.txt | tail -8
