# Round 6, call 5: the image side one batch further ahead (PoseStream lead) against round 5's order, alternating on one box; the pipelined e2e tests; a kernel timeline of each.
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r06c5; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
( time python -m pytest tests/test_gpu_e2e.py tests/test_gpu_bench_contract.py tests/test_gpu_select.py -q -x -k "streamed or pipelined or tail_on" 2>&1 | tail -n 8 ) > $O/pipeline_tests.log 2>&1
B="python -W ignore bench.py --steps 20 --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0"
run() { n=$1; shift; env "$@" SIXDGS_BENCH_DUMP_POSES=1 $B > $O/bench_$n.json 2> $O/bench_$n.err; }
for rep in 1 2 3; do
  run r5order_$rep SIXDGS_POSE_STREAM_LEAD=0
  run lead_$rep SIXDGS_POSE_STREAM_LEAD=1
done
run lead_tail SIXDGS_POSE_STREAM_LEAD=1 SIXDGS_POSE_STREAM_TAIL=1
env SIXDGS_BENCH_DUMP_POSES=1 $B --no-pipeline --b8-steps 0 > $O/bench_nopipe.json 2> $O/bench_nopipe.err
python -W ignore bench.py --config cfg2 --steps 30 --skip-cpu-baseline > $O/bench_cfg2_lead.json 2> $O/bench_cfg2_lead.err
SIXDGS_POSE_STREAM_LEAD=0 python -W ignore bench.py --config cfg2 --steps 30 --skip-cpu-baseline > $O/bench_cfg2_r5order.json 2> $O/bench_cfg2_r5order.err
python -W ignore bench.py --config cfg3 --steps 6 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_cfg3_lead.json 2> $O/bench_cfg3_lead.err
SIXDGS_POSE_STREAM_LEAD=0 python -W ignore bench.py --config cfg3 --steps 6 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 > $O/bench_cfg3_r5order.json 2> $O/bench_cfg3_r5order.err
python - <<PY
import json, glob
import numpy as np
ref = None
for f in ["$O/bench_nopipe.json"] + sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        if ref is None: ref = d.get("poses_last_step")
        b8 = d.get("headline_b8") or {}
        print(f.split("bench_")[1][:-5].ljust(16), d["value"], d["ms_per_step"], "med", d["median_step"]["ms"], "sweep", d["roofline"].get("avg_launch_ms"), "b8", b8.get("value"), b8.get("ms_per_step"),
              "same poses", (np.array_equal(np.asarray(d["poses_last_step"]), np.asarray(ref)) if "poses_last_step" in d and "cfg" not in f else None))
    except Exception as e:
        print(f, "failed", e, open(f.replace(".json", ".err")).read()[-800:])
PY
# timelines
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  SIXDGS_POSE_STREAM_LEAD=$v timeout 600 rocprofv3 --kernel-trace -d $O/trace_lead$v -o trace -- python -W ignore $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0 > $O/trace_lead$v.json 2> $O/trace_lead$v.err
  DB=$(find $O/trace_lead$v -name "*.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB $O/timeline_lead$v.md k_solve_pose 3 > /dev/null 2>&1
  rm -rf $O/trace_lead$v
done
tail -n 5 $O/pipeline_tests.log
