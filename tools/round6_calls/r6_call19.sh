# Round 6, call 19: smoke + the whole GPU suite on the tree with the image-side kernels (durations).
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r06c19; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
( time python -m pytest tests/ -x -q -m gpu --durations=15 2>&1 | tail -n 45 ) > $O/gpu_suite.log 2>&1
tail -n 3 $O/smoke.log; tail -n 32 $O/gpu_suite.log
