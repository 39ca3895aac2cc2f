# round 5, session o: the batch as two staggered sample groups on two queues (tools/experiments/two_groups.py) against the one-engine step
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r5o; mkdir -p $O
timeout 600 python tools/experiments/two_groups.py > $O/two_groups.jsonl 2> $O/two_groups.err; cat $O/two_groups.jsonl; tail -5 $O/two_groups.err
