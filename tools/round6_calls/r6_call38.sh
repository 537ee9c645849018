cd $GRAFT_REPO_ROOT; export SIXDGS_RANDOM_BACKBONE=1
for lead in 0 1; do SIXDGS_DEBUG_LEAD=1 SIXDGS_POSE_STREAM_LEAD=$lead python -W ignore bench.py --steps 10 --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0 2>gpurun_out/dbg_$lead.err >/dev/null; grep -E "per scoring|Error|error|Traceback" -A3 gpurun_out/dbg_$lead.err | tail -8; done
