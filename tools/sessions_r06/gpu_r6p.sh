cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r6p; mkdir -p $O
V=pytorchdeeplearing_amd/lib/variants/libsegengine_b444.so
SEGENGINE_LIB=$V timeout 300 python tools/bench_conv3x_cfgs.py "4,12,128,128:7,44,45,51,52,53,54,55" "4,6,256,256:45,7,51,52,53,54,55" "4,24,64,64:3,51,54" "1,20,128,128:7,51,52,53,54,55" "1,10,256,256:45,51,53,55" "2,16,128,128:7,51,52,53,54" "2,8,256,256:45,51,53,55" > $O/standalone.jsonl 2> $O/standalone.err
cat $O/standalone.jsonl; tail -3 $O/standalone.err
AB="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --roofline-steps 0"
G='"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*'
for i in 1 2 3; do
for cfg in "SEG_NOP=1" "SEG_C3X_MAP=128:128:12=51" "SEG_C3X_MAP=128:128:12=52" "SEG_C3X_MAP=128:128:12=54" "SEG_C3X_MAP=128:128:12=51,256:256:6=53" "SEG_C3X_MAP=128:128:12=53,256:256:6=53"; do
  echo "== $cfg ($i)" >> $O/ab.log; env SEGENGINE_LIB=$V $cfg timeout 200 $AB 2>/dev/null | grep -o "$G" >> $O/ab.log
done; done
cat $O/ab.log
