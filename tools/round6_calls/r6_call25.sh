# Round 6, call 25: per-layer durations of the ray-MLP chain on the round-6 tree (for DESIGN 4d's arithmetic).
cd $GRAFT_REPO_ROOT; bash tools/trace_keys.sh r06c25 2>&1 | tail -n 16; cat gpurun_out/r06c25/time_keys.log | tail -n 3
