# Round 6, call 2: CU-mask bit layout; the split select path (tests); headline A/B of the pipeline variants on ONE box; VRAM wipe-on-release probe.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06c2; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
python -W ignore tools/probe_cumask.py > $O/probe_cumask.log 2>&1
( time python -m pytest tests/test_gpu_select.py -q -x 2>&1 | tail -8 ) > $O/select_tests.log 2>&1
( time python -m pytest tests/test_gpu_e2e.py tests/test_gpu_bench_contract.py -q -x -k "streamed or pipelined" 2>&1 | tail -8 ) > $O/pipeline_tests.log 2>&1
B="python -W ignore bench.py --steps 20 --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0"
run() { # name, env...
  n=$1; shift
  env "$@" SIXDGS_BENCH_DUMP_POSES=1 $B > $O/bench_$n.json 2> $O/bench_$n.err
}
for rep in 1 2; do
  run r5order_$rep SIXDGS_POSE_STREAM_TAIL=0
  run tail_$rep SIXDGS_POSE_STREAM_TAIL=1
  run mask4_$rep SIXDGS_SWEEP_CU_MASK=4
  run mask8_$rep SIXDGS_SWEEP_CU_MASK=8
  run mask1x8_$rep SIXDGS_SWEEP_CU_MASK=1x8
done
env SIXDGS_BENCH_DUMP_POSES=1 $B --no-pipeline --b8-steps 0 > $O/bench_nopipe.json 2> $O/bench_nopipe.err
# VRAM wipe-on-release: a process that touches 200 GB exits; the next process's first big allocation right behind it, and 10 s later
python - <<PY > $O/wipe_probe.log 2>&1
import torch, time
x = torch.empty(200 * 10**9, dtype=torch.uint8, device="cuda"); x.zero_(); torch.cuda.synchronize(); print("held and touched 200 GB", flush=True)
PY
python -W ignore tools/probe_alloc.py --stage 0 >> $O/wipe_probe.log 2>&1
sleep 10
python -W ignore tools/probe_alloc.py --stage 0 >> $O/wipe_probe.log 2>&1
cat $O/probe_cumask.log; tail -4 $O/select_tests.log $O/pipeline_tests.log; grep -v amdgpu.ids $O/wipe_probe.log
python - <<PY
import json, glob
ref = None
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        p = d.get("poses_last_step")
        if "nopipe" in f: ref = p
        b8 = d.get("headline_b8") or {}
        print(f.split("bench_")[1][:-5].ljust(12), d["value"], d["ms_per_step"], "med", d["median_step"]["ms"], "sweep", d["roofline"].get("avg_launch_ms"), "b8", b8.get("value"), b8.get("ms_per_step"), b8.get("sweep_avg_launch_ms"), d["config"]["pipeline"][:0])
    except Exception as e:
        print(f, "failed", e, open(f.replace(".json", ".err")).read()[-600:])
import numpy as np
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("bench_")[1][:-5].ljust(12), "poses identical to --no-pipeline:", ref is not None and np.array_equal(np.asarray(d["poses_last_step"]), np.asarray(ref)))
    except Exception as e: pass
PY
