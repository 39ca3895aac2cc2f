# round 5, session g: finer phase trace of conv3x_kernel (8 stamps per workgroup)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r5g; mkdir -p $O
timeout 200 python tools/trace_conv3x.py 2> $O/conv3x_phase_trace.log > /dev/null; grep "conv3x trace" $O/conv3x_phase_trace.log | awk 'NR%3==0'
