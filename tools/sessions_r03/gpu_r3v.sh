# round 3, session v: fork / tail policy of the weight-gradient stream re-swept on the new kernels
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3v; mkdir -p $O
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
r=d["roofline"]; print("$tag", d["value"], d["ms_per_step"], "host", d["host_enqueue_ms_per_step"], r["ms_per_step"], r["runner_up"])
PY
}
{
run base_1 A=1
run batch2 SEG_FORK_BATCH=2
run batch4 SEG_FORK_BATCH=4
run batch6 SEG_FORK_BATCH=6
run heavy64 SEG_FORK_HEAVY_MB=64
run heavy4 SEG_FORK_HEAVY_MB=4
run tail1 SEG_TAIL_WGRADS=1
run tail2 SEG_TAIL_WGRADS=2
run streams2 SEG_WGRAD_STREAMS=2
run flush0 SEG_FLUSH_LATE=0
run base_2 A=1
} 2>&1 | tee $O/ab.log
