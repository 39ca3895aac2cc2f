# round 2, GPU session H: igemm two-slice stages, head/loss kernels, pack grid, tail weight gradients on the main stream
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops.py tests/test_engine.py tests/test_parallel.py tests/test_stemx.py tests/test_conv3x.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r2h_tests.log
cat gpurun_out/r2h_tests.log
rm -f gpurun_out/r2h_ab.log
for cfg in "SEG_X=0" "SEG_IGEMM_KS=1" "SEG_PACK_WGS=64" "SEG_TAIL_WGRADS=0" "SEG_TAIL_WGRADS=2" "SEG_TAIL_WGRADS=3" "SEG_HEAD_BWD_WGS=512" "SEG_HEAD_BWD_WGS=2048"; do
  echo "== $cfg" >> gpurun_out/r2h_ab.log
  env $cfg timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --roofline-steps 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*' >> gpurun_out/r2h_ab.log
done
cat gpurun_out/r2h_ab.log
rm -rf gpurun_out/trace
timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --roofline-steps 0 > gpurun_out/trace_run.log 2>&1
T=$(find gpurun_out/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $T > gpurun_out/r2h_trace_timeline.txt 2>&1
rm -rf gpurun_out/trace
head -12 gpurun_out/r2h_trace_timeline.txt
SEG_FULLSIZE_REPORT=gpurun_out/r2h_fullsize_report.txt timeout 600 python -m pytest tests/test_fullsize.py -m gpu -q -k low_precision 2>&1 | tail -5
cat gpurun_out/r2h_fullsize_report.txt
