"""Multi-GPU layer: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The pose loop has no cross-image state (pose_estimation/test.py:46-302), so query images shard
across ranks with NO collective on the data path:
  1. rank 0 broadcasts the Gaussian arrays (236 B per Gaussian: 118 MB at N = 500 k -- far cheaper
     than the 36 B x R rays or 1536 B x R keys derived from them) and the scorer weights;
  2. every rank re-emits the rays and rebuilds its key cache locally (deterministic kernels);
  3. images are split into contiguous blocks, one per rank;
  4. rank 0 gathers c2w[B,4,4] (+ status) once per batch -- a few hundred bytes per image.
The helpers are plain tensor plumbing and also run on CPU tensors with the gloo backend (CI tests).
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

_SCENE_FIELDS = ("_xyz", "_scaling", "_rotation", "_features_dc", "_features_rest", "_opacity")


def init_from_env(backend: Optional[str] = None, set_device: bool = True) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from RANK / WORLD_SIZE / LOCAL_RANK; initialises the process group when
    WORLD_SIZE > 1 (rendezvous through MASTER_ADDR / MASTER_PORT, 127.0.0.1 on one node)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl" and set_device:
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of `n` items for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_scene(scene, src: int = 0, device=None):
    """Rank `src` holds the scene; everyone returns a scene with identical arrays.  One broadcast of the
    shapes, then one per attribute array (xGMI point-to-point: 118 MB at 500 k Gaussians)."""
    from .scene import GaussianScene

    if not is_dist():
        return scene
    rank = dist.get_rank()
    dev = device if device is not None else (scene.device if scene is not None else "cpu")
    meta = torch.zeros(2 + len(_SCENE_FIELDS) * 3, dtype=torch.int64, device=dev)
    if rank == src:
        vals = [scene.max_sh_degree, scene.active_sh_degree]
        for f in _SCENE_FIELDS:
            t = getattr(scene, f)
            shp = list(t.shape) + [1] * (3 - t.dim())
            vals += shp[:3]
        meta = torch.tensor(vals, dtype=torch.int64, device=dev)
    dist.broadcast(meta, src)
    m = meta.tolist()
    out = scene if rank == src else GaussianScene(int(m[0]))
    out.active_sh_degree = int(m[1])
    for i, f in enumerate(_SCENE_FIELDS):
        shp = m[2 + 3 * i: 5 + 3 * i]
        if rank == src:
            t = getattr(scene, f).to(dev).contiguous()
        else:
            nd = {"_xyz": 2, "_scaling": 2, "_rotation": 2, "_opacity": 2}.get(f, 3)
            t = torch.empty(shp[:nd], dtype=torch.float32, device=dev)
        dist.broadcast(t, src)
        setattr(out, f, t)
    return out


def broadcast_module(module: torch.nn.Module, src: int = 0):
    """Same weights on every rank (the reference loads one id_module.th; here rank `src` owns it)."""
    if not is_dist():
        return module
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)
    return module


def gather_poses(c2w: torch.Tensor, status: Optional[torch.Tensor] = None, dst: int = 0):
    """Per-rank c2w [b_i,4,4] (b_i may differ) -> on `dst`: concatenation in rank order; None elsewhere."""
    if not is_dist():
        return c2w, status
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([c2w.shape[0]], dtype=torch.int64, device=c2w.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    mx = max(counts) if counts else 0
    payload = torch.zeros(mx, 17, dtype=torch.float32, device=c2w.device)
    payload[: c2w.shape[0], :16] = c2w.reshape(-1, 16)
    if status is not None:
        payload[: c2w.shape[0], 16] = status.float()
    bufs = [torch.zeros_like(payload) for _ in range(world)]
    dist.all_gather(bufs, payload)
    if rank != dst:
        return None, None
    allp = torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
    return allp[:, :16].reshape(-1, 4, 4), allp[:, 16].to(torch.int32)


def barrier():
    if is_dist():
        dist.barrier()


def max_over_ranks(x: float, device) -> float:
    if not is_dist():
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
