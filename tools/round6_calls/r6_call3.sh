# Round 6, call 3: the idle-rank SIGABRT of the 8-ranks-on-one-GPU rehearsal.  40 runs of the evaluation sweep at 8 ranks, every rank with
# faulthandler (all threads), C++ stack traces, HIP error logging and core files; a failing run's cores go through rocgdb.
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r06c3; mkdir -p $O
N=${N_RUNS:-40}
python - <<PY
import importlib, os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
os.environ.setdefault("SIXDGS_RANDOM_BACKBONE", "1")
pkg = importlib.import_module("6dgs_amd"); syn = importlib.import_module("6dgs_amd.synthetic")
from test_gpu_e2e import _write_experiment
root = "/tmp/sweep8"
srcs = syn.write_dataset_fixtures(os.path.join(root, "data"), 1, n_views=34, width=64, height=48)
_write_experiment(root, syn, pkg, "mip_360_room_aa11", srcs["colmap_txt"], 3000, 4)
_write_experiment(root, syn, pkg, "mip_360_garden_bb22", srcs["colmap_bin"], 2500, 5)
_write_experiment(root, syn, pkg, "mip_360_stump_cc33", srcs["colmap_txt"], 2000, 6)
PY
ulimit -c 6000000 2>/dev/null; echo "core limit: $(ulimit -c); pattern: $(cat /proc/sys/kernel/core_pattern)" > $O/summary.log
fails=0
for i in $(seq 1 $N); do
  W=/tmp/sweep8/run_$i; mkdir -p $W; cd $W
  t0=$(date +%s.%N)
  PYTHONFAULTHANDLER=1 TORCH_SHOW_CPP_STACKTRACES=1 AMD_LOG_LEVEL=1 SIXDGS_RANDOM_BACKBONE=1 OMP_NUM_THREADS=2 SIXDGS_DIST_BACKEND=gloo SIXDGS_FORCE_DEVICE=0 \
    PYTHONPATH=$GRAFT_REPO_ROOT timeout 300 python -W ignore -X faulthandler -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $((29600+i)) --tee 3 --log-dir $W/logs \
    $GRAFT_REPO_ROOT/pretrain_eval_attention.py --exp_path /tmp/sweep8/output --out_path $W/res.json --data_type mip360 --skip_train --batch_size 3 --max_ellipsoids -1 > $W/run.log 2>&1
  rc=$?
  t1=$(date +%s.%N)
  echo "run $i rc=$rc $(echo "$t1 - $t0" | bc) s" >> $O/summary.log
  if [ $rc -ne 0 ]; then
    fails=$((fails+1))
    cp $W/run.log $O/fail_run_$i.log
    grep -n -i "abort\|terminate\|core dumped\|Traceback\|Fatal Python\|signal\|HSA_STATUS\|hipError\|:0:\|what()" $W/run.log | head -60 > $O/fail_run_$i.grep
    for c in $(ls $W/core* 2>/dev/null | head -3); do
      echo "== $c" >> $O/fail_run_$i.bt
      timeout 120 /opt/rocm/bin/rocgdb -batch -ex "info threads" -ex "thread apply all bt 40" $(which python3) $c >> $O/fail_run_$i.bt 2>&1
    done
    ls -la $W >> $O/fail_run_$i.grep
    find $W/logs -name "*.log" -size +0 | head -40 | while read f; do echo "== $f"; tail -60 $f; done > $O/fail_run_${i}_ranklogs.txt 2>&1
  fi
  rm -rf $W/core* 2>/dev/null
  cd $GRAFT_REPO_ROOT
done
echo "failures: $fails of $N" >> $O/summary.log
cat $O/summary.log | tail -50
