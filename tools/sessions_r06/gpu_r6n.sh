# round 6, session n: GPU legs of the tests added after the final measurement call (bucketed step and graph replay with the round-6 fusions forced on at small sizes)
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r6n; mkdir -p $O
timeout 900 python -m pytest tests/test_engine.py tests/test_parallel.py -m gpu -q -k "round_6_fusions or graph_replay or bucketed or rccl" > $O/tests.log 2>&1; tail -4 $O/tests.log
SEG_VACT=2 timeout 900 python -m pytest tests/test_engine.py -m gpu -q -k "parity_lowp_gpu or parity_f32_gpu or step_riders or op_ranges or train_steps" > $O/tests_vact2.log 2>&1; tail -4 $O/tests_vact2.log
