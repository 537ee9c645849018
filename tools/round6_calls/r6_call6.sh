# Round 6, call 6: smoke + the whole GPU suite on the pruned tree (durations), then the default bench line.
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r06c6; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
( time python -m pytest tests/ -x -q -m gpu --durations=25 2>&1 | tail -n 60 ) > $O/gpu_suite.log 2>&1
python -W ignore bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -n 3 $O/smoke.log; tail -n 45 $O/gpu_suite.log; head -c 1500 $O/bench_default.json
