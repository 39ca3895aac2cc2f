# round 3, session o: the fp64 yardstick for the f32 gradients at full size
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r3o; mkdir -p $O
SEG_FULLSIZE_REPORT=$O/fp64.txt timeout 900 python -m pytest tests/test_fullsize.py -m gpu -x -q -k "fp64_oracle or full_size_f32" -s 2>&1 | grep -E "f32 vs fp64|worst gradient|passed|failed|Error|assert" | head -12
cat $O/fp64.txt
