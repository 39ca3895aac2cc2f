# round 3, session r: stemx coefficient loads behind the prefetch, single-channel ingest, clDice tile kernels with batched loads
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3r; mkdir -p $O
timeout 900 python -m pytest tests/test_stemx.py tests/test_cldice.py tests/test_engine.py -m gpu -x -q 2>&1 | tail -2 | tee $O/tests.log
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
r=d["roofline"]; print("$tag", d["value"], d["ms_per_step"], "host", d["host_enqueue_ms_per_step"], r["ms_per_step"], r["runner_up"])
PY
}
V=$PWD/pytorchdeeplearing_amd/lib/variants
{
run new_1 A=1
run prev_1 SEGENGINE_LIB=$V/libsegengine_prev.so
run new_2 A=1
run w3total256 SEG_W3_TOTAL=256
run new_3 A=1
} 2>&1 | tee $O/ab.log
timeout 300 python tools/bench_configs.py 2>/dev/null | cut -c1-260 | tee $O/configs.jsonl
timeout 200 python tools/bench_cldice.py 2>/dev/null | tail -8 | tee $O/cldice.log
rm -rf gpurun_out/trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --roofline-steps 0 --condition-seconds 0.2 --launch stream > $O/trace_run.log 2>&1
CSV=$(find gpurun_out/trace -name "*kernel_trace.csv" | head -1)
if [ -n "$CSV" ]; then python tools/trace_gaps.py $CSV > $O/trace_timeline.txt 2>&1; fi
rm -rf gpurun_out/trace
head -5 $O/trace_timeline.txt
