# Round 6, call 4: (a) pipeline A/B with the image + tail streams CONFINED to the CUs a masked sweep leaves out; (b) the 8-rank rehearsal 40 more times with one
# hardware queue per process (GPU_MAX_HW_QUEUES=1: no oversubscription of the device's queue slots).
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r06c4; mkdir -p $O
export SIXDGS_RANDOM_BACKBONE=1
B="python -W ignore bench.py --steps 20 --warmup 2 --skip-cpu-baseline --skip-reference-mode --l32-steps 0 --b8-steps 0"
run() { n=$1; shift; env "$@" SIXDGS_BENCH_DUMP_POSES=1 $B > $O/bench_$n.json 2> $O/bench_$n.err; }
for rep in 1 2; do
  run r5order_$rep SIXDGS_POSE_STREAM_TAIL=0
  run mask4x1_$rep SIXDGS_SWEEP_CU_MASK=4x1
  run mask4x2_$rep SIXDGS_SWEEP_CU_MASK=4x2
  run mask4x4_$rep SIXDGS_SWEEP_CU_MASK=4x4
  run mask4x8_$rep SIXDGS_SWEEP_CU_MASK=4x8
done
run mask4x2_unconfined SIXDGS_SWEEP_CU_MASK=4x2 SIXDGS_SIDE_STREAMS_UNMASKED=1
python - <<PY
import json, glob
import numpy as np
ref = None
rows = []
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        if ref is None: ref = d["poses_last_step"]
        print(f.split("bench_")[1][:-5].ljust(20), d["value"], d["ms_per_step"], "med", d["median_step"]["ms"], "sweep", d["roofline"].get("avg_launch_ms"),
              "same poses", np.array_equal(np.asarray(d["poses_last_step"]), np.asarray(ref)))
    except Exception as e:
        print(f, "failed", e, open(f.replace(".json", ".err")).read()[-800:])
PY
# ---- (b)
N=${N_RUNS:-40}
python - <<PY
import importlib, os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
pkg = importlib.import_module("6dgs_amd"); syn = importlib.import_module("6dgs_amd.synthetic")
from test_gpu_e2e import _write_experiment
root = "/tmp/sweep8"
srcs = syn.write_dataset_fixtures(os.path.join(root, "data"), 1, n_views=34, width=64, height=48)
_write_experiment(root, syn, pkg, "mip_360_room_aa11", srcs["colmap_txt"], 3000, 4)
_write_experiment(root, syn, pkg, "mip_360_garden_bb22", srcs["colmap_bin"], 2500, 5)
_write_experiment(root, syn, pkg, "mip_360_stump_cc33", srcs["colmap_txt"], 2000, 6)
PY
fails=0
for i in $(seq 1 $N); do
  W=/tmp/sweep8/run_$i; mkdir -p $W; cd $W
  t0=$(date +%s)
  PYTHONFAULTHANDLER=1 AMD_LOG_LEVEL=1 OMP_NUM_THREADS=2 SIXDGS_DIST_BACKEND=gloo SIXDGS_FORCE_DEVICE=0 \
    PYTHONPATH=$GRAFT_REPO_ROOT timeout 300 python -W ignore -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $((29700+i)) --tee 3 --log-dir $W/logs \
    $GRAFT_REPO_ROOT/pretrain_eval_attention.py --exp_path /tmp/sweep8/output --out_path $W/res.json --data_type mip360 --skip_train --batch_size 3 --max_ellipsoids -1 > $W/run.log 2>&1
  rc=$?
  echo "run $i rc=$rc $(( $(date +%s) - t0 )) s" >> $O/summary.log
  if [ $rc -ne 0 ]; then
    fails=$((fails+1)); cp $W/run.log $O/fail_run_$i.log
  fi
  rm -f $W/gpucore* 2>/dev/null
  cd $GRAFT_REPO_ROOT
done
echo "failures: $fails of $N (GPU_MAX_HW_QUEUES=1 in every rank)" >> $O/summary.log
tail -n 12 $O/summary.log
