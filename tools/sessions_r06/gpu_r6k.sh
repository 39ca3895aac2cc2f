# round 6, session k: the halo-conv form of the GroupNorm-backward sums alone: SEG_RQ_FUSE=1 (both forms) against 2 (streaming-conv form only) and 0, alternating in one call
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r6k; mkdir -p $O
python -c "from pytorchdeeplearing_amd import _capi; print(_capi.product_library().build_info())" > $O/build.txt 2>&1; cat $O/build.txt
AB="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --roofline-steps 0"
G='"value": [0-9.]*, "unit": "volumes/s", "n_gpus": 1, "steps": 30, "warmup": 5, "ms_per_step": [0-9.]*'
for i in 1 2 3 4; do
  for c in 1 2; do
    echo "== SEG_RQ_FUSE=$c ($i)" >> $O/rq_ab.log; SEG_RQ_FUSE=$c timeout 200 $AB 2>/dev/null | grep -o "$G" >> $O/rq_ab.log
  done
done
cat $O/rq_ab.log
for c in 1 2; do SEG_RQ_FUSE=$c SEG_BENCH_ONLY=C4,C5 timeout 300 python tools/bench_configs.py > $O/configs_rq$c.jsonl 2> $O/configs_rq$c.err; done
cut -c1-100 $O/configs_rq1.jsonl $O/configs_rq2.jsonl
