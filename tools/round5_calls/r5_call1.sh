# Round 5, call 1: k_proj folded into layer 4 -- parity (new test + the tests that touch the chain) and the chain's A/B timing.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fold.py tests/test_gpu_parity.py tests/test_gpu_integration_stub.py -x -q -m gpu -p no:cacheprovider > $O/fold_tests.log 2>&1; tail -5 $O/fold_tests.log
for i in 1 2; do
  python tools/time_keys.py 8388608 2>&1 | grep "planes only" | sed "s/^/fold=1 /" | tee -a $O/chain_fold_ab.log
  SIXDGS_FOLD_KPROJ=0 python tools/time_keys.py 8388608 2>&1 | grep "planes only" | sed "s/^/fold=0 /" | tee -a $O/chain_fold_ab.log
done
(timeout 300 python bench.py --skip-reference-mode > $O/bench_default.json 2> $O/bench_default.err); tail -c 1500 $O/bench_default.json
