# Round 5, call 14: the 8-rank evaluation sweep on the one GPU, five times with every rank's output kept (one run of the suite saw rank 6 abort).
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c14; mkdir -p $O
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
for i in 1 2 3 4 5; do
  SIXDGS_RANDOM_BACKBONE=1 OMP_NUM_THREADS=2 SIXDGS_DIST_BACKEND=gloo SIXDGS_FORCE_DEVICE=0 timeout 300 python -W ignore -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $((29580+i)) --tee 3 \
    pretrain_eval_attention.py --exp_path /tmp/sweep8/output --out_path /tmp/sweep8/res_$i.json --data_type mip360 --skip_train --batch_size 3 --max_ellipsoids -1 > $O/run_$i.log 2>&1
  echo "run $i rc=$?"; grep -n -i "abort\|terminate\|core dumped\|Traceback\|Error" $O/run_$i.log | head -8
done
