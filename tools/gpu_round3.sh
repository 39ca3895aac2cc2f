cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in base occ3 occ4; do
  if [ $v = base ]; then unset SEGENGINE_LIB; else export SEGENGINE_LIB=/root/repo/pytorchdeeplearing_amd/lib/variants/libsegengine_$v.so; fi
  echo "== $v" >> gpurun_out/occ.log
  timeout 200 python tools/bench_conv3.py child >> gpurun_out/occ.log 2>&1
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | cut -c1-330 >> gpurun_out/occ.log
done
cat gpurun_out/occ.log
