# round 3, session z: three waves per SIMD for the 8-tap generic weight-gradient kernel (184 -> 150 VGPRs)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3z; mkdir -p $O
V=$PWD/pytorchdeeplearing_amd/lib/variants
for t in new_1 old_1 new_2 old_2; do
  if [ ${t%_*} = old ]; then export SEGENGINE_LIB=$V/libsegengine_nowaves.so; else unset SEGENGINE_LIB; fi
  timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/$t.json 2> $O/$t.err
  python -c "
import json; d=json.loads(open('$O/$t.json').read().strip().splitlines()[-1]); print('$t', d['value'], d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'])"
done 2>&1 | tee $O/ab.log
