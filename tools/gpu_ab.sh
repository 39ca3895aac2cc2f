cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/ab.log
run() { echo "== $*" >> gpurun_out/ab.log; env "$@" timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --roofline-steps 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*' >> gpurun_out/ab.log; }
run SEG_GNB_MAXROWS=512
run SEG_GNB_MAXROWS=1024
run SEG_GNB_MAXROWS=2048
run SEG_GNB_MAXROWS=4096
run SEG_GNB_MAXROWS=256
run SEG_GNB_MAXROWS=512
cat gpurun_out/ab.log
