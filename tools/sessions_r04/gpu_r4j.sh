# round 4, session j: why does the clDice config measure 11 ms inside bench.py and 6.6 ms in tools/bench_configs.py?
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4j; mkdir -p $O
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --launch stream > $O/bench_stream.json 2> $O/bench_stream.err
python -c "
import json; d=json.loads(open('$O/bench_stream.json').read().strip().splitlines()[-1]); print('launch stream:', d['value'])
for k,v in d['other_configs'].items(): print(k[:70], v.get('ms_per_step'))"
timeout 200 python - <<'PY'
import os, sys, time, json
sys.path.insert(0, os.getcwd())
import torch
import bench
dev = torch.device("cuda:0")
print("fresh process, other_configs only:")
for k, v in bench.other_configs(dev).items(): print(k[:70], v.get("ms_per_step"))
PY
