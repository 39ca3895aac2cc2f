# round 3, session c: persistent double-buffered halo convs (conv3p / conv3p16) on hardware: bit-exactness, standalone timing, in-step A/B
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3c; mkdir -p $O
timeout 600 python -m pytest tests/test_conv3x.py tests/test_engine.py -m gpu -x -q 2>&1 | tail -5 > $O/tests.log; cat $O/tests.log
timeout 300 python tools/tune_conv3x.py --sets top --iters 30 > $O/tune_top.jsonl 2> $O/tune.err; tail -3 $O/tune_top.jsonl | cut -c1-300
grep -E '"cfg": (14|17|18|19|24|25|26|28|29),' $O/tune_top.jsonl | cut -c1-260
B="python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline"
run() { tag=$1; shift; env "$@" timeout 200 $B > $O/$tag.json 2> $O/$tag.err; }
run new_a X=1
run old_a SEG_C3X_MAP=32:32:48=14,16:16:96=24
run new_b X=1
run old_b SEG_C3X_MAP=32:32:48=14,16:16:96=24
run only_p32 SEG_C3X_MAP=16:16:96=24
run only_p16 SEG_C3X_MAP=32:32:48=14
run p32_small SEG_C3X_MAP=32:32:48=19
run wgs512 SEG_C3P16_WGS=768
for f in $O/*.json; do echo "$f $(python -c "
import json,sys
try:
    l=json.loads(open('$f').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], 'mfma', (l.get('roofline_mfma') or {}).get('frac'), (l.get('roofline_mfma') or {}).get('avg_launch_us'))
except Exception as ex: print('ERR', ex)
")"; done
rm -rf gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o step -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --roofline-steps 0 --condition-seconds 0.2 > $O/prof_run.log 2>&1
DB=$(find gpurun_out/prof -name "*.db" | head -1)
if [ -n "$DB" ]; then python profiles/summarize_rocpd.py $DB 50 > $O/kernel_stats.txt 2>&1; fi
rm -rf gpurun_out/prof
head -24 $O/kernel_stats.txt
