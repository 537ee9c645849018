# Round 5, call 2: token packing -- the select tests (packed path) and the sweep's time against the token count.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_select.py tests/test_gpu_full_size.py -x -q -m gpu -p no:cacheprovider > $O/select_tests.log 2>&1; tail -8 $O/select_tests.log
for T in 256 192 128 64; do python tools/time_sweep.py 4 32000000 5 --tokens $T 2>&1 | tail -1 | tee -a $O/time_sweep_tokens.log; done
for T in 128 64; do python tools/time_sweep.py 16 32000000 3 --tokens $T 2>&1 | tail -1 | tee -a $O/time_sweep_tokens.log; done
