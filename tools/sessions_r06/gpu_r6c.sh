# round 6, session c: the batched halo weight-gradient reduce (one reduce launch per level visit: 20 -> 8 launches per VNet3d step) against the previous
# binary (variants/libsegengine_prev.so) in one call; rocprofv3 kernel stats of the new binary; the f32 run dtype launch by launch (which of the
# conv3_kernel<float> launches are far from the f32-MFMA peak); C4 (UNet3d 2 x 128^3) kernel stats + timeline
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6c; mkdir -p $O
python -c "from pytorchdeeplearing_amd import _capi; print(_capi.product_library().build_info())" > $O/build.txt 2>&1; cat $O/build.txt
AB="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --roofline-steps 0"
PREV=pytorchdeeplearing_amd/lib/variants/libsegengine_prev.so
for i in 1 2 3; do
  echo "== new $i" >> $O/ab.log; timeout 200 $AB 2>/dev/null | grep -o '"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*' >> $O/ab.log
  echo "== prev $i" >> $O/ab.log; SEGENGINE_LIB=$PREV timeout 200 $AB 2>/dev/null | grep -o '"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*' >> $O/ab.log
done
# diagnostic variant (conv3.hip -DSEG_DIAG): workgroups per halo weight-gradient launch (256 = one per CU is the product's choice) and boxes per workgroup
W3=pytorchdeeplearing_amd/lib/variants/libsegengine_w3diag.so
for cfg in "SEG_W3_TOTAL=256" "SEG_W3_TOTAL=512" "SEG_W3_TOTAL=512 SEG_W3_MINBOX=3" "SEG_W3_TOTAL=1024 SEG_W3_MINBOX=3" "SEG_W3_TOTAL=256 SEG_W3_MINBOX=12"; do
  echo "== $cfg" >> $O/w3total.log
  env SEGENGINE_LIB=$W3 $cfg timeout 200 $AB 2>/dev/null | grep -o '"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*' >> $O/w3total.log
done
SEG_BENCH_ONLY=C4,C5,C2 timeout 300 python tools/bench_configs.py > $O/configs_new.jsonl 2> $O/configs_new.err
SEGENGINE_LIB=$PREV SEG_BENCH_ONLY=C4,C5,C2 timeout 300 python tools/bench_configs.py > $O/configs_prev.jsonl 2> $O/configs_prev.err
DRV="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs"
rm -rf gpurun_out/prof gpurun_out/trace gpurun_out/trace32 gpurun_out/tracec4
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o step -- $DRV --roofline-steps 0 --launch stream > $O/prof_run.log 2>&1
DB=$(find gpurun_out/prof -name "*.db" | head -1)
if [ -n "$DB" ]; then python profiles/summarize_rocpd.py $DB 60 > $O/rocprofv3_kernel_stats.txt 2>&1; fi
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --roofline-steps 0 --condition-seconds 0.2 --launch stream > $O/trace_run.log 2>&1
CSV=$(find gpurun_out/trace -name "*kernel_trace.csv" | head -1)
if [ -n "$CSV" ]; then python tools/trace_gaps.py $CSV > $O/trace_timeline.txt 2>&1; fi
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace32 -o t -- python bench.py --dtype f32 --steps 3 --warmup 2 --no-cpu-baseline --no-other-configs --roofline-steps 0 --condition-seconds 0.2 --launch stream > $O/trace32_run.log 2>&1
CSV=$(find gpurun_out/trace32 -name "*kernel_trace.csv" | head -1)
if [ -n "$CSV" ]; then python tools/trace_gaps.py $CSV > $O/trace_timeline_f32.txt 2>&1; fi
SEG_BENCH_ONLY=C4 SEG_BENCH_NOPROF=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tracec4 -o t -- python tools/bench_configs.py > $O/tracec4_run.log 2>&1
CSV=$(find gpurun_out/tracec4 -name "*kernel_trace.csv" | head -1)
if [ -n "$CSV" ]; then python tools/trace_gaps.py $CSV > $O/trace_timeline_c4.txt 2>&1; fi
rm -rf gpurun_out/prof gpurun_out/trace gpurun_out/trace32 gpurun_out/tracec4
cat $O/ab.log; cat $O/w3total.log; cut -c1-160 $O/configs_new.jsonl $O/configs_prev.jsonl; head -16 $O/rocprofv3_kernel_stats.txt; head -5 $O/trace_timeline.txt; head -5 $O/trace_timeline_f32.txt; head -5 $O/trace_timeline_c4.txt
