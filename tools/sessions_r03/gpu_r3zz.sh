# round 3, last session: the driver's command, three fresh processes of the FINAL binary
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3zz; mkdir -p $O
for i in 1 2 3; do
  timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 $( [ $i -gt 1 ] && echo --no-cpu-baseline ) > $O/bench_driver_cmd_$i.json 2> $O/bench_driver_cmd_$i.err
  cut -c1-180 $O/bench_driver_cmd_$i.json
done
