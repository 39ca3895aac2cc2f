# round 3, session t: virtual-head variants of the GroupNorm backward kernels (the two 96^3 units under the head), 16-channel wgrad3 policy
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3t; mkdir -p $O
timeout 900 python -m pytest tests/test_engine.py -m gpu -x -q > $O/tests_full.log 2>&1; tail -2 $O/tests_full.log | cut -c1-300 | tee $O/tests.log
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
run vh_1 A=1
run novh_1 SEGENGINE_LIB=$V/libsegengine_novh.so
run vh_2 A=1
run novh_2 SEGENGINE_LIB=$V/libsegengine_novh.so
run vh_t16_128 SEG_W3_TOTAL16=128
run vh_t16_256 SEG_W3_TOTAL16=256
run vh_t16_384 SEG_W3_TOTAL16=384
run vh_t16_256b SEG_W3_TOTAL16=256
run vh_3 A=1
} 2>&1 | tee $O/ab.log
