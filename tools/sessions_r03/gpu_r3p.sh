# round 3, session p: loads issued back to back in the streaming GroupNorm kernels, the generic weight-gradient kernel and the streaming
# convs; register prefetch of the next box in wgrad3_kernel.  A/B inside one call: previous binary / only GN / only weight gradients / all.
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3p; mkdir -p $O
timeout 900 python -m pytest tests/test_ops.py tests/test_engine.py tests/test_conv3x.py -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.log
V=pytorchdeeplearing_amd/lib/variants
for rep in 1 2; do
for tag in prev gn_only wg_only new; do
  if [ $tag = new ]; then unset SEGENGINE_LIB; else export SEGENGINE_LIB=$PWD/$V/libsegengine_$tag.so; fi
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${tag}_$rep.json 2> $O/${tag}_$rep.err
  python - <<PY
import json
d=json.loads(open("$O/${tag}_$rep.json").read().strip().splitlines()[-1])
r=d["roofline"]; print("$tag $rep", d["value"], d["ms_per_step"], "host", d["host_enqueue_ms_per_step"], r["kernel"][:24], r["ms_per_step"], r["runner_up"])
PY
done; done 2>&1 | tee $O/ab.log
unset SEGENGINE_LIB
