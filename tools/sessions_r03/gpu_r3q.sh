# round 3, session q: after the back-to-back loads - conv_igemm bias / tap loads in the prologue; launch-policy knobs re-tuned (the
# weight-gradient kernels no longer expose their staging latency, so fewer / more workgroups may win now); kernel trace of the new state
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3q; mkdir -p $O
timeout 600 python -m pytest tests/test_ops.py -m gpu -x -q 2>&1 | tail -2 | tee $O/tests.log
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
r=d["roofline"]; print("$tag", d["value"], d["ms_per_step"], "host", d["host_enqueue_ms_per_step"], r["ms_per_step"], r["runner_up"])
PY
}
{
run base_1 A=1
run w3total256 SEG_W3_TOTAL=256
run w3total384 SEG_W3_TOTAL=384
run w3total768 SEG_W3_TOTAL=768
run w3min4 SEG_W3_MINBOX=4
run w3min9 SEG_W3_MINBOX=9
run w3t16_512 SEG_W3_TOTAL16=512
run w3t16_2048 SEG_W3_TOTAL16=2048
run fold1024 SEG_FOLD_WGS=1024
run fold4096 SEG_FOLD_WGS=4096
run base_2 A=1
} 2>&1 | tee $O/ab.log
rm -rf gpurun_out/trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --roofline-steps 0 --condition-seconds 0.2 --launch stream > $O/trace_run.log 2>&1
CSV=$(find gpurun_out/trace -name "*kernel_trace.csv" | head -1)
if [ -n "$CSV" ]; then python tools/trace_gaps.py $CSV > $O/trace_timeline.txt 2>&1; fi
rm -rf gpurun_out/trace
head -8 $O/trace_timeline.txt; grep -A40 "per-phase kernel totals" $O/trace_timeline.txt | head -70
