# round 3, session w: wgrad3 workgroup-count policy on ONE box for every BASELINE config (256 / 256 against 512 / 1024 and mixes)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3w; mkdir -p $O
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
r=d["roofline"]; print("$tag", d["value"], d["ms_per_step"], "host", d["host_enqueue_ms_per_step"], r["ms_per_step"], r["runner_up"])
PY
}
cfg() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python tools/bench_configs.py 2>/dev/null | grep -v cldice | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print('$tag', d['config'][:28], d['ms_per_step'], d.get('class_ms', {}).get('wgrad3'))
"
}
{
run d256_256_a A=1
run d512_1024_a SEG_W3_TOTAL=512 SEG_W3_TOTAL16=1024
run d256_1024 SEG_W3_TOTAL=256 SEG_W3_TOTAL16=1024
run d512_256 SEG_W3_TOTAL=512 SEG_W3_TOTAL16=256
run d256_256_b A=1
run d512_1024_b SEG_W3_TOTAL=512 SEG_W3_TOTAL16=1024
cfg c256_256 A=1
cfg c512_1024 SEG_W3_TOTAL=512 SEG_W3_TOTAL16=1024
cfg c256_1024 SEG_W3_TOTAL=256 SEG_W3_TOTAL16=1024
cfg c512_256 SEG_W3_TOTAL=512 SEG_W3_TOTAL16=256
} 2>&1 | tee $O/ab.log
