# Round 5, call 3: quarter-granular token packing -- select tests and the sweep's time against the token count; then the whole -m gpu suite.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_select.py -x -q -m gpu -p no:cacheprovider > $O/select_tests.log 2>&1; tail -3 $O/select_tests.log
for T in 256 192 128 64; do python tools/time_sweep.py 4 32000000 5 --tokens $T 2>&1 | tail -1 | tee -a $O/time_sweep_tokens.log; done
for T in 176 100; do python tools/time_sweep.py 16 32000000 3 --tokens $T 2>&1 | tail -1 | tee -a $O/time_sweep_tokens.log; done
timeout 1200 python -m pytest tests -x -q -m gpu --durations=8 -p no:cacheprovider > $O/gpu_suite_full.log 2>&1
grep -v "^E    +" $O/gpu_suite_full.log | tail -16
