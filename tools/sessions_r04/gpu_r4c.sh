# round 4, session c: where does a wgrad3 workgroup spend its time?  (phase sums, diagnostic build)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r4c; mkdir -p $O
for arm in "SEG_W3_BOX16=0" "SEG_W3_BOX16=1" "SEG_W3_CQ=32"; do
  echo "== $arm"; env $arm timeout 120 python tools/trace_wgrad3.py 2>&1 | grep "wgrad3 trace" | awk 'NR%3==0'
done > $O/trace.log 2>&1
cat $O/trace.log
