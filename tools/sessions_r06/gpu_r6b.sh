# round 6, session b: the pruned library (no SEG_EXPERIMENTS, no flag forks / sub-batches / VACT / reduce fold) - bench line, the GPU parity suite, and a
# DIAGNOSTIC variant (-DSEG_DIAG: wrong gradients) that drops weight-gradient launches, to bound what the second queue costs the step:
# SEG_DIAG_NOWGRAD = 0 (none dropped) / 1 (all) / 2 (levels >= 24^3) / 3 (96^3 and 48^3)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r6b; mkdir -p $O
python -c "from pytorchdeeplearing_amd import _capi; print(_capi.product_library().build_info())" > $O/build.txt 2>&1; cat $O/build.txt
DRV="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs"
timeout 300 $DRV > $O/bench_1.json 2> $O/bench_1.err
V=pytorchdeeplearing_amd/lib/variants/libsegengine_diag.so
for m in 0 1 2 3 0 1 2 3; do
  echo "== SEG_DIAG_NOWGRAD=$m" >> $O/nowgrad.log
  SEGENGINE_LIB=$V SEG_DIAG_NOWGRAD=$m timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --roofline-steps 0 2>/dev/null | grep -o '"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": [0-9.]*' >> $O/nowgrad.log
done
timeout 300 $DRV > $O/bench_2.json 2> $O/bench_2.err
SEG_FULLSIZE_REPORT=$O/fullsize_report.txt timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $O/gpu_tests_full.log 2>&1; tail -25 $O/gpu_tests_full.log > $O/gpu_tests.log
cat $O/gpu_tests.log; cat $O/nowgrad.log; for i in 1 2; do cut -c1-200 $O/bench_$i.json; done
