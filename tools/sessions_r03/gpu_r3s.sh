# round 3, session s: GPU tests again (the stemx dummy bias pointer was null in the operator-level statistics mode), weight-gradient
# launch-policy sweep on the prefetching kernels
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3s; mkdir -p $O
timeout 900 python -m pytest tests/test_stemx.py tests/test_cldice.py tests/test_engine.py tests/test_ops.py -m gpu -x -q > $O/tests_full.log 2>&1; tail -4 $O/tests_full.log | cut -c1-300 | tee $O/tests.log
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
run w3t128 SEG_W3_TOTAL=128
run w3t192 SEG_W3_TOTAL=192
run w3t512 SEG_W3_TOTAL=512
run wg1024 SEG_WG_TOTAL=1024
run wg4096 SEG_WG_TOTAL=4096
run w3t16_512 SEG_W3_TOTAL16=512
run w3t16_256 SEG_W3_TOTAL16=256
run base_2 A=1
} 2>&1 | tee $O/ab.log
