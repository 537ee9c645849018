"""Host-side pieces added in round 3 that need no GPU: the ellipsoid blocks of ray-sharded emission, the pose-difference measure of the
bench line's parity_vs_oracle, the device scoping of the ops front end (pass-through without a GPU tensor), the sample strides."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_shard_blocks_partition_the_ellipsoids_on_granule_boundaries():
    sampling = importlib.import_module("6dgs_amd.sampling")
    for n in (0, 1, 255, 256, 257, 10_000, 500_000, 2_000_001):
        for world in (1, 2, 3, 8):
            blocks = [sampling.shard_block(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            for (a0, a1), (b0, b1) in zip(blocks, blocks[1:]):
                assert a1 == b0 and a0 <= a1                                  # contiguous, in rank order, possibly empty at the end
            for lo, hi in blocks:
                assert lo % 256 == 0 or lo == n                               # every block starts on a 256-ellipsoid granule (key-tile boundary)
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) == sizes[0]                                     # equal blocks; what is left goes to the last non-empty rank(s)


def test_pose_delta_is_well_conditioned_at_small_angles():
    bench = importlib.import_module("bench")
    a = np.eye(4)
    for deg in (1e-6, 1e-4, 0.5, 30.0, 179.0):
        t = np.radians(deg)
        b = np.eye(4)
        b[:3, :3] = [[np.cos(t), -np.sin(t), 0], [np.sin(t), np.cos(t), 0], [0, 0, 1]]
        b[:3, 3] = [0.3, -0.4, 1.2]
        rot, tr = bench._pose_delta(a, b)
        assert abs(rot - deg) <= 1e-9 + 1e-7 * deg and abs(tr - 1.3) < 1e-12
    # fp32-rounded copies of ONE rotation: acos((trace - 1) / 2) would report ~0.02 degrees here
    q, _ = np.linalg.qr(np.random.default_rng(0).standard_normal((3, 3)))
    c = np.eye(4)
    c[:3, :3] = q
    d = c.astype(np.float32).astype(np.float64)
    assert bench._pose_delta(c, d)[0] < 2e-5


def test_ops_front_end_scopes_nothing_without_gpu_tensors_and_refuses_cpu_tensors():
    ops = importlib.import_module("6dgs_amd.ops")
    assert ops._first_device([1, "x", torch.zeros(2), [torch.zeros(1)], None]) is None
    assert ops._first_device(["cpu"]) is None
    with pytest.raises(RuntimeError, match="must live on the GPU"):
        ops.mask_degraded(torch.zeros(4, 3))
    assert getattr(ops._tls, "device", None) is None                                   # the thread-local device is put back


def test_select_sample_stride_follows_the_scene_size():
    ops = importlib.import_module("6dgs_amd.ops")
    for r, stride in ((1 << 20, 16), (15_999_999, 16), (16_000_000, 32), (31_999_999, 32), (32_000_000, 64), (512_000_000, 64)):
        n = r // stride
        i = torch.tensor([0, 1, n // 2, n - 1], dtype=torch.int64)
        ref = i * stride + (((i * 2654435761) & 0xFFFFFFFF) >> 13) % stride
        idx = ops.select_sample_indices.__wrapped__(r, "cpu")
        assert idx.shape[0] == n and torch.equal(idx[i], ref) and int(idx.max()) < r
        assert bool((idx[1:] > idx[:-1]).all())
