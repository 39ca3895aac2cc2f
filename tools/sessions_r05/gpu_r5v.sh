# round 5, session v: in-call A/B of the two misc.hip changes (column sum with <= 512 blocks of atomics; multi-class loss_backward unrolled) - the library linked
# with the previous misc.hip (lib/ab/libsegengine_prevmisc.so) against the current one, C2 / C4 step times interleaved three times (leases differ by 2-3 %)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r5v; mkdir -p $O
for i in 1 2 3; do
  echo "== previous misc.hip" >> $O/ab.log
  SEGENGINE_LIB=$PWD/pytorchdeeplearing_amd/lib/ab/libsegengine_prevmisc.so SEG_BENCH_ONLY=C2,C4 timeout 300 python tools/bench_configs.py 2>/dev/null | cut -c1-100 >> $O/ab.log
  echo "== current" >> $O/ab.log
  SEG_BENCH_ONLY=C2,C4 timeout 300 python tools/bench_configs.py 2>/dev/null | cut -c1-100 >> $O/ab.log
done
cat $O/ab.log
