# round 6, session l: soak of the final binary - long runs (every launch of gn_bwd_coop_kernel polls its partners: a lost word would poison the gradients with NaN and show as
# skipped steps / a non-finite loss), other batch sizes and volume sizes (other slice counts S and chunk counts K of the one-launch GroupNorm backward, other sides of the
# 16 MB floors), the bf16 and f32 run dtypes, graph launch mode; then the GPU parity suite of the final tree
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r6y; mkdir -p $O
python -c "from pytorchdeeplearing_amd import _capi; print(_capi.product_library().build_info())" > $O/build.txt 2>&1; cat $O/build.txt
Q="--no-cpu-baseline --no-other-configs --roofline-steps 0"
F='"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"final_loss": [-0-9.a-zA-Z]*\|"skipped_steps": [0-9]*\|"launch_mode": "[a-z]*"'
run() { echo "== $*" >> $O/soak.log; timeout 600 python bench.py $Q "$@" 2>>$O/soak.err | grep -o "$F" | tr '\n' ' ' >> $O/soak.log; echo >> $O/soak.log; }
run --steps 3000 --warmup 5
run --steps 3000 --warmup 5
run --steps 1000 --warmup 5 --batch 1
run --steps 1000 --warmup 5 --batch 2
run --steps 500 --warmup 5 --batch 8
run --steps 300 --warmup 5 --batch 16
run --steps 500 --warmup 5 --size 64
run --steps 500 --warmup 5 --size 128 --batch 2
run --steps 300 --warmup 5 --size 160 --batch 1 --dtype bf16
run --steps 300 --warmup 5 --dtype bf16
run --steps 100 --warmup 5 --dtype f32
run --steps 500 --warmup 5 --launch graph
cat $O/soak.log
