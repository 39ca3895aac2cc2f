# round 4, session t: why is the flag fork slower in the free-running step although the main queue's gaps are gone?  host enqueue time per mode,
# and the runtime's hipStreamWaitValue32 (SEG_FORK=flag) against an own one-lane polling kernel (SEG_FORK=spin)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4t; mkdir -p $O
for m in event flag spin; do echo "host $m: $(SEG_FORK=$m timeout 120 python tools/host_enqueue.py 2>/dev/null | tail -1)"; done 2>&1 | tee $O/host_enqueue.log
SEG_FORK=spin timeout 300 python -m pytest tests/test_engine.py -x -q -m gpu -k "flag_forks" 2>&1 | tail -2
run() { t=$1; shift
  env "$@" timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --launch stream > $O/$t.json 2> $O/$t.err
  python -c "
import json; d=json.loads(open('$O/$t.json').read().strip().splitlines()[-1]); print('$t', d['value'], d['ms_per_step'])"
}
{
run spin_1 SEG_FORK=spin
run event_1 SEG_FORK=event
run flag_1 SEG_FORK=flag
run spin_2 SEG_FORK=spin
run event_2 SEG_FORK=event
} 2>&1 | tee $O/ab.log
