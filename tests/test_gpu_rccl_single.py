"""RCCL on the one GPU of the test box (VERDICT r3 #5a): a process group of ONE rank with backend `nccl` (= RCCL on ROCm), created by
`distributed.init_from_env` under SIXDGS_DIST_SINGLE=1, so that every collective of `6dgs_amd/distributed.py` -- broadcast_scene,
broadcast_module, gather_poses, gather_results, merge_row_stats, merge_topk, kth_largest_of_union, all_counts, agree, ranks_seen,
max_over_ranks, barrier -- runs through RCCL on DEVICE tensors instead of being skipped (world size 1 without the switch) or moving CPU
tensors over gloo (the two-rank tests).  The multi-GPU node the driver has then runs the same code with world size N.
Each test compares with the run of the same command without a process group: same poses / same file."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TORCHRUN = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1"]


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SIXDGS_RANDOM_BACKBONE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **kw)
    return env


def _json_line(stdout):
    lines = [l for l in stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_bench_over_rccl_with_one_rank_gives_the_poses_of_the_plain_run():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    args = ["bench.py", "--gpus", "1", "--gaussians", "40000", "--steps", "2", "--warmup", "1", "--skip-cpu-baseline", "--skip-reference-mode", "--l32-steps", "0"]
    plain = subprocess.run([sys.executable, "-W", "ignore", *args], cwd=ROOT, env=_env(SIXDGS_BENCH_DUMP_POSES="1"), capture_output=True, text=True, timeout=400)
    assert plain.returncode == 0, plain.stderr[-2000:]
    rccl = subprocess.run([sys.executable, "-W", "ignore", *TORCHRUN, "--master-port", "29561", *args], cwd=ROOT,
                          env=_env(SIXDGS_BENCH_DUMP_POSES="1", SIXDGS_DIST_SINGLE="1", SIXDGS_BENCH_BACKEND="nccl"), capture_output=True, text=True, timeout=400)
    assert rccl.returncode == 0, rccl.stderr[-2000:]
    a, b = _json_line(plain.stdout), _json_line(rccl.stdout)
    assert a["backend"] == "none" and a["ranks_seen"] == 1
    assert b["backend"] == "nccl" and b["ranks_seen"] == 1 and b["n_gpus"] == 1       # ranks_seen: an all-reduce of ones on device memory over RCCL
    pa, pb = np.asarray(a["poses_last_step"]), np.asarray(b["poses_last_step"])
    assert pa.shape == pb.shape == (8, 4, 4)
    assert np.array_equal(pa, pb)                                                     # scene + weights went through RCCL broadcasts, poses through the gather


@pytest.mark.timeout(900)
def test_ray_sharded_bench_and_checks_over_rccl_with_one_rank():
    """--parallelism ray at one rank: merge_row_stats / the g_t and key-norm all-reduces / kth_largest_of_union / merge_topk / the selected
    rays' all-reduce all run over RCCL; tools/ray_shard_check.py compares the ray-sharded scorers (two-pass and select, incl. a forced
    fall-back) with the single-GPU ones."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = _env(SIXDGS_DIST_SINGLE="1", SIXDGS_BENCH_BACKEND="nccl", SIXDGS_DIST_BACKEND="nccl")
    p = subprocess.run([sys.executable, "-W", "ignore", *TORCHRUN, "--master-port", "29563", "bench.py", "--gpus", "1", "--parallelism", "ray", "--gaussians", "40000",
                        "--batch", "3", "--steps", "2", "--warmup", "1", "--skip-cpu-baseline", "--skip-reference-mode", "--l32-steps", "0"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _json_line(p.stdout)
    assert d["backend"] == "nccl" and d["scaling"] == "strong" and d["config"]["parallelism"].startswith("ray-sharded x1") and d["value"] > 0
    for extra, rays in (([], "20037"), (["--select"], "300037")):
        p = subprocess.run([sys.executable, "-W", "ignore", *TORCHRUN, "--master-port", "29565", "tools/ray_shard_check.py", "--backend", "nccl", "--rays", rays, *extra],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
        assert p.returncode == 0, p.stderr[-2000:]
        r = _json_line(p.stdout)
        assert r["ok"] and r["ranks"][0]["world"] == 1, r
        assert r.get("backend") == "nccl", r


@pytest.mark.timeout(900)
def test_evaluation_sweep_over_rccl_with_one_rank_writes_the_same_file(tmp_path):
    """pretrain_eval_attention.py under torch.distributed.run with one nccl rank: broadcast_scene, broadcast_module, the emission seed,
    agree() behind every stage and gather_results go through RCCL; results.json equals the plain run's."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import importlib
    pkg = importlib.import_module("6dgs_amd")
    syn = importlib.import_module("6dgs_amd.synthetic")
    from test_gpu_e2e import _write_experiment
    root = str(tmp_path)
    srcs = syn.write_dataset_fixtures(os.path.join(root, "data"), 1, n_views=26, width=64, height=48)
    _write_experiment(root, syn, pkg, "mip_360_room_aa11", srcs["colmap_txt"], 3000, 4)
    outs = []
    for name, launcher, env in (("plain", [], _env()), ("rccl", [*TORCHRUN, "--master-port", "29567"], _env(SIXDGS_DIST_SINGLE="1", SIXDGS_DIST_BACKEND="nccl"))):
        out = os.path.join(root, f"res_{name}.json")
        p = subprocess.run([sys.executable, "-W", "ignore", *launcher, os.path.join(ROOT, "pretrain_eval_attention.py"), "--exp_path", os.path.join(root, "output"),
                            "--out_path", out, "--data_type", "mip360", "--skip_train", "--batch_size", "3", "--max_ellipsoids", "-1"], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=500)
        assert p.returncode == 0, p.stderr[-2000:]
        if name == "rccl":
            assert "backend nccl" in (p.stdout + p.stderr)
        outs.append(json.load(open(out)))
    assert len(outs[0]) == 4 and [r["frame_id"] for r in outs[1]] == [0, 1, 2, 3]
    for a, b in zip(*outs):
        assert a["gt_c2w"] == b["gt_c2w"] and a["pred_c2w"] == b["pred_c2w"]
