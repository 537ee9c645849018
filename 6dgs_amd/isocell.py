"""pose_estimation/isocell.py, module surface (SURVEY.md 8 a8/a9 and what VERDICT r2 listed as the last unbuilt corner of the file).

    isocell_distribution(ray_target, dtype, device, N0=1, isrand=-1)   isocell.py:6-84
    rotate_isocell(isocell_directions, normal)                          isocell.py:171-222
    group_by_360_isocell(dirs, ray_target, N0=3)                        isocell.py:87-140
    get_dirs_group_idx(dirs, ray_target, N0=3)                          isocell.py:143-154

The two functions on the hot path (the deterministic distribution, isrand = -1, and the Rodrigues rotation) are the HIP kernels
behind `ops.isocell_distribution` / `ops.rotate_isocell`.  The rest is never called by the reference; it is restated here on the host
(a few tensor operations on <= a few hundred directions), pinned by golden g13:

* isrand in {1, 2, 3, 4}: the reference adds a per-RING offset th0 (shape [n]) to a per-CELL vector (shape [N0 n^2]) (isocell.py:24,44),
  which only broadcasts for n = 1: every target with more than one ring raises RuntimeError there, and so it does here.  For n = 1 the
  draws are made from torch's CPU generator in the reference's order (rand(1); then per mode rand / randn of shape [1, N0]), so a
  `torch.manual_seed` before the call reproduces the reference's CPU result bit for bit.
* the grouping helpers keep the reference's quirks: ring ids are clamped at N0 (not at n - 1), the group key adds the cell id twice, and
  the group with the largest key is never returned (`range(key.max())`).
"""
from __future__ import annotations

import math
from typing import List

import torch

from . import ops


def _rings(ray_target: int, n0: int) -> int:
    return int(math.ceil(math.sqrt(ray_target / n0)))


def isocell_distribution(ray_target, dtype=torch.float32, device="cuda", N0: int = 1, isrand: int = -1, int_dtype=torch.int64) -> torch.Tensor:
    if isrand == -1:
        if dtype != torch.float32:
            raise RuntimeError("6dgs_amd: the iso-cell directions are computed in fp32 (the reference's dtype on this path)")
        return ops.isocell_distribution(int(ray_target), int(N0), device=device)
    # Every other value of isrand takes the reference's random start angle; 1..4 add their per-cell jitter, anything else centres the cells in
    # radius and angle (isocell.py:61-63, the reference's else-branch -- ADVICE r3: this used to raise ValueError here).  All draws come from
    # the generator of `device`, in the reference's order (isocell.py:24,47-60 pass device=device): a CPU seed reproduces the reference bit for
    # bit (golden g13), a CUDA seed reproduces torch's device generator as the reference would use it.
    n = _rings(ray_target, N0)
    u0 = torch.rand(1, dtype=dtype, device=device)        # (the reference draws th0 before it looks at isrand)
    if n != 1:
        raise RuntimeError(f"The size of tensor a ({n}) must match the size of tensor b ({N0 * n * n}) at non-singleton dimension 0 "
                           "(isocell.py:44 adds a per-ring offset to a per-cell vector: the random modes exist for one ring only)")
    cells = N0                                            # one ring of N0 cells, radius step dR = 1
    dth = 2 * math.pi / torch.tensor([float(cells)], dtype=dtype, device=device)   # (a tensor, so that the fp32 rounding is the reference's)
    start = u0 * dth + torch.arange(cells, dtype=dtype, device=device) * dth           # [cells]
    radius = torch.ones(cells, dtype=dtype, device=device)

    def half_gauss():
        return (1 + torch.randn(1, cells, dtype=dtype, device=device) / 6.5) / 2

    if isrand == 1:
        radius = radius - torch.rand(1, cells, dtype=dtype, device=device) * 1.0
        theta = start + torch.rand(1, cells, dtype=dtype, device=device) * dth
    elif isrand == 2:
        radius = radius - torch.rand(1, cells, dtype=dtype, device=device) * 1.0
        theta = start + dth / 2
    elif isrand == 3:
        radius = radius - half_gauss() * 1.0
        theta = start + half_gauss() * dth / 2
    elif isrand == 4:
        radius = radius - half_gauss() * 1.0
        theta = start + dth / 2
    else:
        radius = radius - 1.0 / 2                         # (1-D here: the reference's result is [cells, 3] in this branch, [1, 3 cells] in modes 1-4)
        theta = start + dth / 2
    x, y = radius * torch.cos(theta), radius * torch.sin(theta)
    z = torch.real(torch.sqrt(1 - torch.square(x.to(torch.complex64)) - torch.square(y.to(torch.complex64))))
    return torch.column_stack([x, y, z])


def rotate_isocell(isocell_directions: torch.Tensor, normal: torch.Tensor) -> torch.Tensor:
    return ops.rotate_isocell(isocell_directions, normal)


def group_by_360_isocell(dirs: torch.Tensor, ray_target: int, N0: int = 3, int_dtype=torch.int64):
    """-> (hemisphere [D] (1 = below the z = 0 plane), ring id [D], cell id within the ring [D]) of unit directions `dirs` [D,3]."""
    n = _rings(ray_target, N0)
    step = 1 / n
    per_ring = N0 * (2 * torch.arange(1, n + 1, dtype=dirs.dtype, device=dirs.device) - 1)
    below = dirs[:, 2] < 0
    flat = dirs.clone()
    flat[:, 2] = dirs[:, 2] - dirs[:, 2] * 1.0            # the component along z removed: exactly 0, as the reference asserts
    rad = torch.linalg.norm(flat, dim=-1)
    ring = torch.floor_divide(rad, step).to(int_dtype).clamp(min=0, max=N0)
    ang_c, ang_s = torch.arccos(flat[:, 0] / rad), torch.arcsin(flat[:, 1] / rad)
    theta = torch.where(ang_s >= 0, ang_c, -ang_c) + math.pi
    width = (2 * math.pi) / per_ring[ring]
    cell = torch.floor_divide(theta, width).to(int_dtype)
    return below.to(int_dtype), ring, cell


def get_dirs_group_idx(dirs: torch.Tensor, ray_target: int, N0: int = 3, int_dtype=torch.int64) -> List[torch.Tensor]:
    """Index sets of the directions that share a (hemisphere, ring, cell) key, in ascending key order, empty keys dropped."""
    below, ring, cell = group_by_360_isocell(dirs, ray_target, N0=N0, int_dtype=int_dtype)
    key = (below * (ring.max() + 1) + ring) * (cell.max() + 1) + cell
    key = key + cell                                       # (the reference adds the cell id a second time)
    order = torch.argsort(key, stable=True)                # ascending key, ascending index inside a key
    sorted_key = key[order]
    keep = sorted_key < key.max()                          # `range(key.max())` never reaches the largest key
    order, sorted_key = order[keep], sorted_key[keep]
    if order.numel() == 0:
        return []
    _, counts = torch.unique_consecutive(sorted_key, return_counts=True)
    return list(torch.split(order.to(int_dtype), counts.tolist()))
