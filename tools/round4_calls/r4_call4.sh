# Round 4, fourth GPU call: the test files a global -k filter skipped in the third call, and the whole-step hipGraph under --hip-trace.
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4c4; mkdir -p $O
timeout 900 python -m pytest -q -m gpu -p no:cacheprovider --durations=8 tests/test_gpu_select.py tests/test_gpu_integration_stub.py tests/test_isocell_module.py tests/test_gpu_rccl_single.py tests/test_gpu_bench_contract.py 2>&1 | grep -v "^E    +" | tail -40 > $O/tests_a.log; tail -22 $O/tests_a.log
timeout 400 python -m pytest -q -m gpu -p no:cacheprovider -x -s "tests/test_gpu_configs.py::test_headline_500k_x64_scores_and_top100_against_the_oracle" 2>&1 | grep -E "passed|failed|rror|key parity|\[headline|assert" | tail -12 > $O/tests_b.log; cat $O/tests_b.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --hip-trace -d $O/trace_graph -o trace -- python $GRAFT_REPO_ROOT/bench.py --config cfg2 --graph --steps 10 --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_cfg2_graph.json 2> $O/trace_graph.err
DB=$(find $O/trace_graph -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB $O/cfg2_graph_replay_timeline.md k_solve_pose 8 2>&1 | tail -3
python - <<PY
import sqlite3
db = sqlite3.connect("$DB"); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
print("tables/views:", tabs[:60])
for v in ("regions", "top"):
    if v in tabs:
        print(v, [r[1] for r in cur.execute(f"pragma table_info({v})")])
PY
rm -rf $O/trace_graph
grep -n "busy\|HIP API" -A 16 $O/cfg2_graph_replay_timeline.md | tail -40
