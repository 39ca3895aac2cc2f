cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r6o; mkdir -p $O
AB="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --roofline-steps 0"
G='"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*'
V=tools/experiments/libsegengine_diag.so
for i in 1 2 3; do
for cfg in "SEG_DIAG_COOP_WGS_BIG=256" "SEG_DIAG_COOP_WGS_BIG=512" "SEG_DIAG_COOP_WGS_BIG=1024" "SEG_DIAG_COOP_KB=4096"; do
  echo "== $cfg ($i)" >> $O/coop.log; env SEGENGINE_LIB=$V $cfg timeout 200 $AB 2>/dev/null | grep -o "$G" >> $O/coop.log
done; done
cat $O/coop.log
