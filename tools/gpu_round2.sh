set -x
cd /root/repo; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/bench_prepost.py > gpurun_out/prepost2.jsonl 2> gpurun_out/prepost2.err
timeout 300 python tools/bench_cldice.py 4,1,96,96,96 2,1,96,96,96 1,1,96,96,96 1,1,192,192,192 > gpurun_out/cldice2.jsonl 2> gpurun_out/cldice2.err
rm -rf gpurun_out/prof_cldice
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cldice -o cl -- python tools/bench_cldice.py 4,1,96,96,96 > gpurun_out/cldice_prof.log 2>&1
find gpurun_out/prof_cldice -name "*kernel_stats*" | head -1 | xargs -I{} head -20 {} > gpurun_out/cldice_kernel_stats.csv
find gpurun_out/prof_cldice -name "*kernel_trace*" -size +10M -delete
cat gpurun_out/prepost2.jsonl gpurun_out/cldice2.jsonl; cat gpurun_out/cldice_kernel_stats.csv | cut -c1-200
