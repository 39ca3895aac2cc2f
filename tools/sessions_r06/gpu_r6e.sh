# round 6, session e: the input block's instruction diet (stemx.hip: range-test-free tile loop for boxes inside the volume, unsigned box arithmetic carried over from the
# prefetch, shifted halo copies written without a range test, 4 x 8 x 16 boxes for the two forward passes) against the previous binary, in one call; the one-launch
# GroupNorm backward limited to tensors <= 4 MB (diag: up to 8 MB and 512 workgroups); parity tests of what changed
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6e; mkdir -p $O
python -c "from pytorchdeeplearing_amd import _capi; print(_capi.product_library().build_info())" > $O/build.txt 2>&1; cat $O/build.txt
timeout 900 python -m pytest tests/test_stemx.py tests/test_engine.py -m gpu -q -x > $O/tests.log 2>&1; tail -5 $O/tests.log
AB="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --roofline-steps 0"
G='"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*'
PREV=tools/experiments/libsegengine_prev.so
for i in 1 2 3; do
  echo "== new ($i)" >> $O/ab.log; timeout 200 $AB 2>/dev/null | grep -o "$G" >> $O/ab.log
  echo "== prev ($i)" >> $O/ab.log; SEGENGINE_LIB=$PREV timeout 200 $AB 2>/dev/null | grep -o "$G" >> $O/ab.log
done
cat $O/ab.log
V=tools/experiments/libsegengine_diag.so
for cfg in "SEG_DIAG_SX_FWD_BOX4=1" "SEG_DIAG_SX_FWD_BOX4=0" "SEG_DIAG_COOP_KB=8192" "SEG_DIAG_COOP_KB=8192 SEG_DIAG_COOP_WGS=512" "SEG_DIAG_COOP_WGS=512" "SEG_DIAG_SX_FWD_BOX4=1" "SEG_DIAG_SX_FWD_BOX4=0"; do
  echo "== $cfg" >> $O/diag.log; env SEGENGINE_LIB=$V $cfg timeout 200 $AB 2>/dev/null | grep -o "$G" >> $O/diag.log
done
cat $O/diag.log
SEG_BENCH_ONLY=C4,C5,C2 timeout 300 python tools/bench_configs.py > $O/configs_new.jsonl 2> $O/configs_new.err
SEGENGINE_LIB=$PREV SEG_BENCH_ONLY=C4,C5,C2 timeout 300 python tools/bench_configs.py > $O/configs_prev.jsonl 2> $O/configs_prev.err
cut -c1-260 $O/configs_new.jsonl $O/configs_prev.jsonl
rm -rf gpurun_out/trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --roofline-steps 0 --condition-seconds 0.2 --launch stream > $O/trace_run.log 2>&1
CSV=$(find gpurun_out/trace -name "*kernel_trace.csv" | head -1)
if [ -n "$CSV" ]; then python tools/trace_gaps.py $CSV > $O/trace_timeline.txt 2>&1; fi
rm -rf gpurun_out/trace
head -4 $O/trace_timeline.txt; grep -n "stemx_kernel" $O/trace_timeline.txt | head -12
