# round 5, session p: the three fresh-process lines of the driver's command again with bench.py's PMC row filter fixed (roofline_2 / _3 `traffic` of the mangled-name kernels)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r5p; mkdir -p $O
DRV="python bench.py --gpus 1 --steps 20 --warmup 5"
for i in 1 2 3; do
  timeout 500 $DRV $( [ $i -gt 1 ] && echo --no-cpu-baseline --no-other-configs ) > $O/bench_driver_cmd_$i.json 2> $O/bench_driver_cmd_$i.err
  cut -c1-200 $O/bench_driver_cmd_$i.json
done
