# Round 5, call 20: the whole -m gpu suite on the final tree (no -x: every test runs).
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c20; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1500 python -m pytest tests -q -m gpu --durations=8 -p no:cacheprovider > $O/gpu_suite_full.log 2>&1
grep -v "^E    +" $O/gpu_suite_full.log | tail -14
