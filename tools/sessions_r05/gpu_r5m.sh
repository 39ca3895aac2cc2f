# round 5, session m: GroupNorm-backward reduce folded into the conv3x data-gradient epilogue - equivalence tests on the GPU, then an in-call
# A/B of the driver's command: previous binary (853a7e666d76, lib/ab/) | this binary with SEG_GN_RFUSE=1 (default) | this binary with SEG_GN_RFUSE=0, three rounds
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r5m; mkdir -p $O
timeout 600 python -m pytest tests/test_engine.py -m gpu -x -q -k "reduce_folded or conv3x_path" > $O/tests.log 2>&1; tail -2 $O/tests.log
DRV="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs"
SHOW='import sys,json; d=json.loads(sys.stdin.read()); print(d["build"], d["value"], d["ms_per_step"], {k:(v["ms_per_step"],v["launches_per_step"]) for k,v in d["kernel_families"].items() if "halo" in k.lower() or "groupnorm" in k.lower() or "gn" in k.lower()})'
for r in 1 2 3; do
  echo "== prev" >> $O/ab.log;  SEGENGINE_LIB=$PWD/pytorchdeeplearing_amd/lib/ab/libsegengine_853a.so timeout 300 $DRV 2>/dev/null | python -c "$SHOW" >> $O/ab.log
  echo "== fold=1" >> $O/ab.log; SEG_GN_RFUSE=1 timeout 300 $DRV 2>/dev/null | python -c "$SHOW" >> $O/ab.log
  echo "== fold=0" >> $O/ab.log; SEG_GN_RFUSE=0 timeout 300 $DRV 2>/dev/null | python -c "$SHOW" >> $O/ab.log
done
cat $O/ab.log
# the other BASELINE configs, fold on / off (one line each)
for f in 1 0; do echo "== other configs fold=$f" >> $O/configs.log; SEG_GN_RFUSE=$f timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], {k:v.get("ms_per_step") for k,v in d.get("other_configs",{}).items()})' >> $O/configs.log; done
cat $O/configs.log
