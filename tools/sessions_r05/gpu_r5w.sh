# round 5, session w: the weight re-layout split three ways (SEG_PACK_SPLIT=2: only the layouts of the two finest encoder levels on the caller's stream) -
# parity tests on the GPU, in-call A/B of the driver's command against the two-way split (SEG_PACK_SPLIT=1), four rounds
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r5w; mkdir -p $O
python -c "from pytorchdeeplearing_amd import _capi; print(_capi.product_library().build_info())" > $O/build.txt 2>&1; cat $O/build.txt
timeout 900 python -m pytest tests/test_engine.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
DRV="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --roofline-steps 0"
for i in 1 2 3 4; do
  for m in 1 2; do echo -n "SEG_PACK_SPLIT=$m: " >> $O/ab.log; SEG_PACK_SPLIT=$m timeout 300 $DRV 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])' >> $O/ab.log; done
done
cat $O/ab.log
