# last call of the round: smoke, three fresh-process bench lines with the driver's command (PMC summary of the same build in profiles/), the GPU parity suite of the final tree
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r6x; mkdir -p $O
python -c "from pytorchdeeplearing_amd import _capi; print(_capi.product_library().build_info())" > $O/build.txt 2>&1; cat $O/build.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
for i in 1 2 3; do
  timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 $( [ $i -gt 1 ] && echo --no-cpu-baseline --no-other-configs ) > $O/bench_driver_cmd_$i.json 2> $O/bench_driver_cmd_$i.err
  cut -c1-200 $O/bench_driver_cmd_$i.json
done
SEG_FULLSIZE_REPORT=$O/fullsize_report.txt timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $O/gpu_tests_full.log 2>&1; tail -25 $O/gpu_tests_full.log > $O/gpu_tests.log; tail -3 $O/gpu_tests.log
