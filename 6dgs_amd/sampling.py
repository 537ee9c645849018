"""generate_all_possible_rays -- drop-in for pose_estimation/sampling.py:127-267.

Reference flow (once per scene): validity mask over all N Gaussians (always at 50 target cells,
sampling.py:140-144) -> randperm subsample of min(1000, valid) ellipsoids (:145-149) -> 20-NN
normals (:151-166) -> quadricell surface cells, rotation, hemisphere mask, ray origin/direction
(:168-196) -> SH colour per ray from the source Gaussian (:225-251).

Here every stage is a HIP kernel reading the raw parameter arrays (activations fused); the only
host sync is the ragged ray count, where the reference syncs too.  Extensions beyond the reference
(keyword-only, defaults reproduce it):
  max_ellipsoids   1000 = the reference's cap; -1 = emit from EVERY valid Gaussian
  emitter          "quadricell" (live path) | "isocell" (pose_estimation/isocell.py: K directions per
                   ellipsoid, what BASELINE.json's "64 / 256 isocell rays per ellipsoid" describes)
  perm             the subsample permutation (otherwise torch.randperm, as the reference)
  shard            (rank, world): emit only this rank's contiguous block of the selected ellipsoids (ray-sharded scenes,
                   SURVEY 8(e) fallback).  The normals still come from the 20-NN search over ALL selected ellipsoids, so the
                   concatenation of the shards' rays in rank order IS the unsharded ray set.  Blocks are whole multiples of 256
                   ellipsoids (every shard's first ray then sits on a key-tile boundary of the unsharded scene).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops


def shard_block(n: int, rank: int, world: int, granule: int = 256):
    """[lo, hi) of rank's block of n ellipsoids: equal blocks rounded up to whole granules, the last rank(s) take what is left."""
    per = -(-n // world)
    per = -(-per // granule) * granule
    return min(rank * per, n), min((rank + 1) * per, n)


@torch.no_grad()
def generate_all_possible_rays(
    model,
    num_viewdirs_per_chunk=10240,  # kept for signature parity; the fused kernels do not chunk
    sample_quadricell_targets=50,
    *,
    max_ellipsoids: int = 1000,
    emitter: str = "quadricell",
    rays_per_ellipsoid: int = 64,
    perm: Optional[torch.Tensor] = None,
    k_neighbors: int = 20,
    return_src: bool = False,
    shard: Optional[tuple] = None,
):
    xyz, log_scale, rot = model._xyz, model._scaling, model._rotation
    if not xyz.is_cuda:
        raise RuntimeError("6dgs_amd.generate_all_possible_rays needs the scene on the GPU (no CPU fallback)")
    dev = xyz.device
    mask_valid = ops.mask_degraded(log_scale, 50)
    valid_ids = torch.nonzero(mask_valid)[:, 0]
    valid_num = int(valid_ids.shape[0])
    if max_ellipsoids is None or max_ellipsoids < 0:
        if perm is None:
            sel = valid_ids
        else:
            sel = valid_ids[perm.to(dev)]
    else:
        if perm is None:
            perm = torch.randperm(valid_num, dtype=torch.long, device=dev)
        sel = valid_ids[perm.to(dev)[: min(max_ellipsoids, valid_num)]]
    sel = sel.contiguous()
    centres = xyz[sel].contiguous()
    normals = ops.normals_knn(centres, centres, k_neighbors) if sel.shape[0] else torch.empty(0, 3, device=dev)
    if shard is not None:
        lo, hi = shard_block(int(sel.shape[0]), int(shard[0]), int(shard[1]))
        sel, normals = sel[lo:hi].contiguous(), normals[lo:hi].contiguous()
    if emitter == "quadricell":
        ori, dr, rgb, src, _ = ops.emit_quadricell(xyz, log_scale, rot, model._features_dc, model._features_rest,
                                                   int(model.active_sh_degree), sel, normals, int(sample_quadricell_targets))
    elif emitter == "isocell":
        dirs = ops.isocell_distribution(int(rays_per_ellipsoid), 1, device=dev)
        ori, dr, rgb, src = ops.emit_isocell(xyz, log_scale, rot, model._features_dc, model._features_rest,
                                             int(model.active_sh_degree), sel, normals, dirs, want_src=return_src)
    else:
        raise ValueError(f"unknown emitter {emitter!r}")
    if return_src:
        return ori, dr, rgb, src
    return ori, dr, rgb
