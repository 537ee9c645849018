"""world_size-2 gloo tests of the multi-GPU plumbing (scene broadcast, module broadcast, image sharding,
pose gather) on CPU tensors -- the same code path bench.py runs over RCCL."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port))
        dd = importlib.import_module("6dgs_amd.distributed")
        pkg = importlib.import_module("6dgs_amd")
        syn = importlib.import_module("6dgs_amd.synthetic")
        r, w, _ = dd.init_from_env("gloo")
        assert (r, w) == (rank, world) and dd.is_dist()
        # 1. scene broadcast from rank 0
        ref = syn.make_scene(257, 3)
        scene = pkg.GaussianScene.from_dict(ref, device="cpu") if rank == 0 else None
        scene = dd.broadcast_scene(scene, 0, device="cpu")
        assert len(scene) == 257 and scene.active_sh_degree == 3
        for f, k in (("_xyz", "xyz"), ("_scaling", "log_scale"), ("_rotation", "rot"), ("_features_dc", "f_dc"),
                     ("_features_rest", "f_rest")):
            assert np.array_equal(getattr(scene, f).numpy(), ref[k]), f
        # 2. module broadcast
        lin = torch.nn.Linear(5, 3)
        with torch.no_grad():
            lin.weight.fill_(float(rank + 1))
        dd.broadcast_module(lin, 0)
        assert float(lin.weight[0, 0]) == 1.0
        # 3. image sharding + gather of ragged per-rank pose blocks
        n_img = 7
        lo, hi = dd.shard_range(n_img, rank, world)
        c2w = torch.stack([torch.eye(4) * (i + 1) for i in range(lo, hi)]) if hi > lo else torch.zeros(0, 4, 4)
        status = torch.arange(lo, hi, dtype=torch.int32)
        allp, alls = dd.gather_poses(c2w, status, 0)
        if rank == 0:
            assert allp.shape == (n_img, 4, 4)
            assert [float(allp[i, 0, 0]) for i in range(n_img)] == [float(i + 1) for i in range(n_img)]
            assert alls.tolist() == list(range(n_img))
        else:
            assert allp is None
        # 4. timing reduction
        assert dd.max_over_ranks(float(rank), "cpu") == float(world - 1)
        dd.barrier()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
        raise e


@pytest.mark.timeout(180)
def test_two_rank_gloo_pipeline_plumbing():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
