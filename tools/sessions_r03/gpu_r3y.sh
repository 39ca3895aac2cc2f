# round 3, session y: scatter-form streaming conv with the bias / GroupNorm sums folded over the taps (Cout = 16): 273 -> 122 VGPRs
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3y; mkdir -p $O
timeout 600 python -m pytest tests/test_ops.py -m gpu -x -q -k "conv" 2>&1 | tail -2 | tee $O/tests.log
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
run fold_1 A=1
run nofold_1 SEGENGINE_LIB=$V/libsegengine_nofold.so
run fold_2 A=1
run nofold_2 SEGENGINE_LIB=$V/libsegengine_nofold.so
} 2>&1 | tee $O/ab.log
