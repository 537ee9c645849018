"""The evaluation sweep -- drop-in for the reference's entry point pretrain_eval_attention.py:

    python -m sixdgs_amd.pretrain_eval_attention --exp_path <3DGS output dir> --out_path results.json --data_type mip360

for every trained scene under --exp_path (`<prefix>..._<id>/point_cloud/iteration_N/point_cloud.ply` + `cfg_args`):
PLY -> device arrays, cameras of the scene's dataset, train-or-load `id_module.th`, ray emission, the ground-truth-ray pass and the
true inference pass of test_pose_estimation, one results list for all scenes written as JSON (schema test.py:290-302).  Same
function names, arguments and per-scene RuntimeError handling as the reference (pretrain_eval_attention.py:20-248); behind them the
MI355X path of this package.

Multi-GPU (north_star: queries shard over independent test images): started under `torch.distributed.run`, every rank loads the
scene (rank 0 reads the PLY and the checkpoint and broadcasts them over RCCL; ray emission and the key cache are rebuilt locally,
deterministically), takes a contiguous block of the scene's test views, and rank 0 gathers the per-view result dicts in view order
before writing the file.  Training, when no checkpoint exists, runs on rank 0 only and is broadcast.

Beyond the reference (keyword-only / extra flags, defaults reproduce it): --emitter / --max_ellipsoids / --rays_per_ellipsoid select
full-scene emission (every Gaussian) instead of the reference's 1000-ellipsoid subsample; --skip_train evaluates random-init
weights when no checkpoint exists (smoke runs); --n_iterations shortens training.
"""
from __future__ import annotations

import argparse
import json
import os
import traceback
from functools import partial
from typing import Any, List, Optional

import numpy as np
import torch

from . import distributed as dd
from .datasets import dotdict, get_checkpoint_arguments, load_data, parse_exp_dir
from .distance_based_loss import DistanceBasedScoreLoss
from .identification_module import IdentificationModule
from .sampling import generate_all_possible_rays
from .scene import GaussianScene
from .test import test_pose_estimation
from .train import train_id_module

EMISSION = dict(max_ellipsoids=1000, emitter="quadricell", rays_per_ellipsoid=64)      # the reference's emission (sampling.py:145-149,175-196)


def parse_args(argv=None):
    """pose_estimation/opt.py:4-29 (plain argparse: configargparse is not a dependency) + this build's extras."""
    ap = argparse.ArgumentParser(add_help=True)
    ap.add_argument("--exp_path", type=str, required=True, help="experiment directory")
    ap.add_argument("--out_path", type=str, required=True, help="results file (json)")
    ap.add_argument("--data_type", type=str, choices=["blender", "mip360", "tankstemple", "cambridge_landmark", "all"], default="all",
                    help="the type of data to validate")
    ap.add_argument("--emitter", choices=["quadricell", "isocell"], default="quadricell")
    ap.add_argument("--max_ellipsoids", type=int, default=1000, help="ellipsoids that emit rays (1000 = the reference; -1 = every valid Gaussian)")
    ap.add_argument("--rays_per_ellipsoid", type=int, default=64, help="iso-cell emitter only")
    ap.add_argument("--n_iterations", type=int, default=1500, help="training iterations when no id_module.th exists")
    ap.add_argument("--skip_train", action="store_true", help="never train: evaluate the checkpoint, or random-init weights when there is none")
    ap.add_argument("--batch_size", type=int, default=16, help="query images per scorer launch")
    ap.add_argument("--arena_gb", type=float, default=0.0,
                    help="carve the big per-scene buffers (key planes, select workspace, chain workspace) from ONE device buffer of this many GB allocated once "
                         "(ops.Arena) instead of asking the allocator scene by scene: a sweep over scenes of growing size otherwise pays a 100-200 GB hipMalloc "
                         "per scene.  -1: everything free but 56 GB; 0 (default): off")
    return ap.parse_known_args(argv)


def load_model(checkpoint_path, device, sh_degrees=3):
    """pretrain_eval_attention.py:20-28: the 3DGS checkpoint as device arrays (no autograd state to switch off here)."""
    return GaussianScene.load_ply(checkpoint_path, sh_degree=sh_degrees, device=device)


def explore_model(model, **emission):
    """pretrain_eval_attention.py:163-169."""
    kw = dict(EMISSION)
    kw.update(emission)
    return generate_all_possible_rays(model, sample_quadricell_targets=50, **kw)


def pretrain_single_object(checkpoint_filepath: str, checkpoint_args: "dotdict[str, Any]", exp_dir_filepath: str, object_id: str, category_name: str,
                           starting_seed: int, lock_backbone: bool = True, device: str = "cuda", *, emission: Optional[dict] = None,
                           n_iterations: int = 1500, skip_train: bool = False, batch_size: int = 16, backbone: Optional[torch.nn.Module] = None):
    """pretrain_eval_attention.py:31-160 for one scene; returns the result dicts of the inference pass for ALL test views (rank 0;
    other ranks return their own block).

    Multi-rank: the scene is walked in STAGES.  A stage is rank-local work that may fail (`dd.agree`: the ranks all-reduce a
    failure flag behind it and leave the scene TOGETHER on a RuntimeError of any of them); the collectives (scene / weight
    broadcast, seed, result gather) sit between the stages, where every rank is known to arrive.  Rank 0 trains a missing
    checkpoint inside a stage while the others wait in that stage's all-reduce (on the group with the long timeout: agree(long_wait=True),
    distributed.init_from_env)."""
    torch.manual_seed(starting_seed)
    print("data_path: ", checkpoint_args.source_path)
    emission = dict(EMISSION, **(emission or {}))
    ckpt_path = os.path.join(exp_dir_filepath, "id_module.th")

    def stage_load():
        scene = load_model(checkpoint_filepath, device, sh_degrees=checkpoint_args.sh_degree if checkpoint_args.sh_degree is not None else 3) \
            if dd.rank() == 0 else None
        if checkpoint_args.fps_sampling is None:
            checkpoint_args.fps_sampling = -1
        info = load_data(checkpoint_args)
        module = IdentificationModule(backbone_type="dino", backbone=backbone).to(device).train()
        start = 0
        if os.path.exists(ckpt_path):
            print("Checkpoint already exist, skip training phase")
            ckpt = torch.load(ckpt_path, map_location=device)
            module.load_state_dict(ckpt["model_state_dict"])
            start = ckpt["epoch"]
        return scene, info, module, start

    scene, scene_info, id_module, start_iterations = dd.agree(stage_load, "load scene, cameras and checkpoint", device)
    gs_model = dd.broadcast_scene(scene, 0, device=device)

    def stage_train():
        if not skip_train and start_iterations < n_iterations and dd.rank() == 0:
            train_id_module(ckpt_path, device, id_module, partial(explore_model, gs_model, **emission), scene_info, object_id, category_name,
                            start_iterations=start_iterations, lock_backbone=lock_backbone, n_iterations=n_iterations)

    dd.agree(stage_train, "train the scorer (rank 0)", device, long_wait=True)
    dd.broadcast_module(id_module, 0)
    id_module.eval()
    id_module.invalidate_caches()
    print("Training complete starting testing phase...")
    if dd.is_dist():         # every rank must emit the SAME rays (the subsample is a torch.randperm): one seed from rank 0's generator
        torch.manual_seed(dd.broadcast_int(int(torch.randint(0, 2**31 - 1, (1,)).item()), 0, device))
    lo, hi = dd.shard_range(len(scene_info.test_cameras), dd.rank(), dd.world())

    def stage_evaluate():
        rays_ori, rays_dirs, rays_rgb = explore_model(gs_model, **emission)
        model_up = torch.from_numpy(np.mean(np.asarray([c.R[:3, 1] for c in scene_info.train_cameras], dtype=np.float32), axis=0)).to(device)
        mine = scene_info.test_cameras[lo:hi]
        print("Testing overfit performances...")
        _, o_t, o_a, o_s, o_r = test_pose_estimation(mine, id_module, rays_ori, rays_dirs, rays_rgb, model_up, sequence_id=object_id,
                                                     category_id=category_name, loss_fn=DistanceBasedScoreLoss(), batch_size=batch_size)
        print("Overfit AVG translation error: ", o_t)
        print("Overfit AVG angular error: ", o_a)
        print("Overfit AVG score error: ", o_s)
        print("Overfit recall: ", o_r)
        print("Testing performances on same points...")
        results, t_t, t_a, t_s, t_r = test_pose_estimation(mine, id_module, rays_ori, rays_dirs, rays_rgb, model_up, sequence_id=object_id,
                                                           category_id=category_name, save=False, save_all=False, batch_size=batch_size)
        for r in results:
            r["frame_id"] += lo                  # frame ids count the scene's test views, not the rank's block
        print("Test AVG translation error: ", t_t)
        print("Test AVG angular error: ", t_a)
        print("Test AVG score error: ", t_s)
        print("Test recall: ", t_r)
        return results

    results = dd.agree(stage_evaluate, "emit rays and evaluate the test views", device)
    return dd.gather_results(results, 0)


def evaluate_single_object_in_blender(checkpoint_filepath: str, checkpoint_args, exp_dir_filepath: str, object_id: str, category_name: str,
                                      starting_seed: int = 55176280, device: str = "cuda", lock_backbone: bool = True, **extras):
    """pretrain_eval_attention.py:172-197."""
    return pretrain_single_object(checkpoint_filepath, checkpoint_args, exp_dir_filepath, object_id, category_name, starting_seed,
                                  device=device, lock_backbone=lock_backbone, **extras)


PREFIXES = {"blender": "synthetic_", "mip360": "mip_360_", "tankstemple": "tt_", "cambridge_landmark": "cl_"}


def main(argv=None, backbone: Optional[torch.nn.Module] = None) -> List[dict]:
    args, _ = parse_args(argv)
    forced = os.environ.get("SIXDGS_FORCE_DEVICE")            # test hook: several ranks on one GPU (gloo)
    rank, world, local = dd.init_from_env(set_device=forced is None)
    if forced is not None:
        local = int(forced)
    out_path_abs = os.path.abspath(args.out_path)
    if rank == 0:
        os.makedirs(os.path.dirname(out_path_abs), exist_ok=True)
    if not torch.cuda.is_available():
        raise RuntimeError("6dgs_amd: the evaluation sweep needs an MI355X (no CPU fallback on the product path)")
    device = f"cuda:{local}"
    torch.cuda.set_device(local)
    dd.warm_long_wait_group(torch.device(device))          # (exists already unless init_from_env left it to us: nccl with a forced device)
    if dd.is_dist():
        seen = dd.ranks_seen(torch.device(device))          # a collective: every rank takes part, rank 0 reports
        if rank == 0:
            print(f"[6dgs_amd] evaluation sweep over {seen} rank(s), backend {dd.backend_name()}")
    results: List[dict] = []
    arena = None
    if args.arena_gb:
        from . import ops
        free_b = torch.cuda.mem_get_info(torch.device(device))[0]
        want = int(free_b - (56 << 30)) if args.arena_gb < 0 else int(args.arena_gb * (1 << 30))
        arena = ops.Arena(max(1 << 28, min(want, int(0.95 * free_b))), torch.device(device))
        ops.set_arena(arena)
        if rank == 0:
            print(f"[6dgs_amd] arena of {arena.capacity / 2**30:.1f} GiB for the per-scene buffers")
    for exp in parse_exp_dir(args.exp_path, PREFIXES.get(args.data_type, "")).values():
        ckpt_args = get_checkpoint_arguments(exp["exp_dir_filepath"])
        try:
            obj = evaluate_single_object_in_blender(
                exp["checkpoint_filepath"], ckpt_args, exp["exp_dir_filepath"], exp["sequence_id"], exp["category_name"], starting_seed=55176280,
                device=device, lock_backbone=True,
                emission=dict(emitter=args.emitter, max_ellipsoids=args.max_ellipsoids, rays_per_ellipsoid=args.rays_per_ellipsoid),
                n_iterations=args.n_iterations, skip_train=args.skip_train, batch_size=args.batch_size, backbone=backbone)
            if rank == 0:
                results.extend(obj)
        except RuntimeError:            # the only exception the reference survives per scene (pretrain_eval_attention.py:243-244)
            traceback.print_exc()
    if arena is not None:
        from . import ops
        ops.set_arena(None)
    if rank == 0:
        print("Saving results")
        with open(out_path_abs, "w") as fh:
            json.dump(results, fh)
    dd.barrier()
    return results


if __name__ == "__main__":
    torch.manual_seed(71170)
    main()
