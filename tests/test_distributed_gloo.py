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
        # 5. a stage that fails on ONE rank is left by BOTH (a rank-local RuntimeError -- OOM, unreadable scene -- used to
        #    send that rank on to the next scene's broadcast while the other still sat in this scene's collectives)
        assert dd.agree(lambda: rank * 10, "fine") == rank * 10

        def stage():
            if rank == 1:
                raise RuntimeError("rank 1 ran out of memory")
            return "done"

        try:
            dd.agree(stage, "evaluate")
            raised = None
        except RuntimeError as e:
            raised = str(e)
        assert raised is not None and (("rank 1 ran out of memory" in raised) if rank == 1 else ("another rank failed during 'evaluate'" in raised))
        t = torch.tensor([float(rank + 1)])
        torch.distributed.broadcast(t, 0)                 # the next scene's first collective pairs up again
        assert float(t) == 1.0
        # 6. an exception that is NOT a RuntimeError -- a loader's FileNotFoundError, a checkpoint's KeyError -- reaches the
        #    all-reduce as well: the failing rank re-raises its own, the peer gets the RuntimeError; and a long-wait stage (rank 0 trains)
        #    runs its all-reduce on the group with the long timeout while the default group keeps the backend's default
        def stage_io():
            if rank == 1:
                raise FileNotFoundError("cameras.json")
            return "loaded"

        try:
            dd.agree(stage_io, "load")
            raised = None
        except (RuntimeError, FileNotFoundError) as e:
            raised = e
        assert isinstance(raised, FileNotFoundError if rank == 1 else RuntimeError), raised
        assert dd.agree(lambda: "trained", "train", long_wait=True) == "trained"
        assert dd.long_wait_group() is dd.long_wait_group()
        torch.distributed.broadcast(t, 0)
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


def _worker_ray_shard(rank, world, port, q):
    """The collectives of the ray-sharded scorer on CPU tensors: statistics merge and candidate merge."""
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port))
        dd = importlib.import_module("6dgs_amd.distributed")
        dd.init_from_env("gloo")
        rng = np.random.default_rng(5)
        B, T, R, K = 3, 256, 1000, 100
        logits = (rng.standard_normal((B, T, R)) * 6).astype(np.float32)
        logits[1, 40:] = -np.inf                     # image 1 has 40 tokens: the other rows never see a ray
        lo, hi = dd.shard_range(R, rank, world)
        loc = logits[:, :, lo:hi].astype(np.float64)
        m = loc.max(axis=2)
        with np.errstate(invalid="ignore"):
            sl = np.where(np.isinf(m), 0.0, np.exp(loc - np.where(np.isinf(m), 0.0, m)[..., None]).sum(axis=2))
        stats = torch.from_numpy(np.stack([m, sl], axis=-1).astype(np.float32))
        g = dd.merge_row_stats(stats).numpy().astype(np.float64)
        M = logits.astype(np.float64).max(axis=2)
        with np.errstate(invalid="ignore"):
            S = np.where(np.isinf(M), 0.0, np.exp(logits.astype(np.float64) - np.where(np.isinf(M), 0.0, M)[..., None]).sum(axis=2))
        assert np.array_equal(g[..., 0], M.astype(np.float32).astype(np.float64))
        assert np.allclose(g[..., 1], S, rtol=1e-6, atol=0)
        # candidates: value ties across ranks resolve to the lower global index; short shards pad with (-1, NaN)
        scores = rng.integers(0, 50, size=(B, R)).astype(np.float32)          # heavy ties
        sc_loc = scores[:, lo:hi]
        order = np.lexsort((np.broadcast_to(np.arange(hi - lo), sc_loc.shape), -sc_loc), axis=1)[:, :K]
        idx = torch.from_numpy(order.astype(np.int64))
        val = torch.from_numpy(np.take_along_axis(sc_loc, order, axis=1))
        if rank == 1:                                                          # pretend this shard has only 7 rays for image 2
            idx[2, 7:] = -1
            val[2, 7:] = float("nan")
        gi, gv = dd.merge_topk(idx, val, lo, K)
        ref = scores.copy()
        lo1, hi1 = dd.shard_range(R, 1, world)
        keep1 = order if rank == 1 else np.lexsort((np.broadcast_to(np.arange(hi1 - lo1), (B, hi1 - lo1)), -scores[:, lo1:hi1]), axis=1)[:, :K]
        mask = np.ones(R, bool)
        mask[lo1:hi1] = False
        mask[lo1 + keep1[2, :7]] = True
        ref[2, ~mask] = -np.inf
        ro = np.lexsort((np.broadcast_to(np.arange(R), ref.shape), -ref), axis=1)[:, :K]
        assert np.array_equal(gi.numpy(), ro), (gi[2, :12], ro[2, :12])
        assert np.array_equal(gv.numpy(), np.take_along_axis(ref, ro, axis=1))
        # ray-sharded SELECT plumbing: U_(k) of the scene from the shards' top-k lists; the selected rays put together from their owners
        u = rng.random((B, R)).astype(np.float32)
        u[0, :] = np.round(u[0, :] * 20) / 20                                  # heavy ties in image 0
        ul = np.sort(u[:, lo:hi], axis=1)[:, ::-1][:, :K].copy()
        if rank == 1:
            ul[2, 5:] = np.nan                                                 # a shard with only 5 rays for image 2
        uk = dd.kth_largest_of_union(torch.from_numpy(ul), K).numpy()
        full = u.copy()
        full[2, lo1:hi1] = -np.inf
        full[2, lo1:lo1 + 5] = np.sort(u[2, lo1:hi1])[::-1][:5]
        assert np.array_equal(uk, np.sort(full, axis=1)[:, ::-1][:, K - 1])
        assert dd.all_counts(hi - lo, "cpu") == [dd.shard_range(R, r_, world)[1] - dd.shard_range(R, r_, world)[0] for r_ in range(world)]
        ori = torch.from_numpy(rng.standard_normal((R, 3)).astype(np.float32))
        dr = torch.from_numpy(rng.standard_normal((R, 3)).astype(np.float32))
        gsel = torch.from_numpy(rng.integers(0, R, size=(B, 17)))
        gsel[1, 3] = -1
        so, sd_ = dd.gather_selected_rays(gsel, ori[lo:hi], dr[lo:hi], lo)
        want_o, want_d = ori[gsel.clamp(min=0)], dr[gsel.clamp(min=0)]
        want_o[1, 3], want_d[1, 3] = 0.0, 0.0
        assert torch.equal(so, want_o) and torch.equal(sd_, want_d)
        dd.barrier()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
        raise e


@pytest.mark.timeout(180)
def test_two_rank_gloo_ray_sharded_merges():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ray_shard, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def _single_worker(port, q):
    try:
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SIXDGS_DIST_SINGLE="1")
        import importlib
        import torch
        dd = importlib.import_module("6dgs_amd.distributed")
        assert not dd.is_dist()
        assert dd.init_from_env("gloo") == (0, 1, 0)
        assert dd.is_dist() and dd.backend_name() == "gloo" and dd.world() == 1 and dd.ranks_seen("cpu") == 1
        assert dd.all_counts(7, "cpu") == [7] and dd.max_over_ranks(2.5, "cpu") == 2.5
        c2w, st = dd.gather_poses(torch.eye(4)[None], torch.tensor([3], dtype=torch.int32), 0)
        assert c2w.shape == (1, 4, 4) and st.tolist() == [3]
        assert dd.gather_results([{"a": 1}]) == [{"a": 1}] and dd.agree(lambda: 5, "x") == 5 and dd.agree(lambda: 6, "y", long_wait=True) == 6
        dd.barrier()
        q.put("ok")
    except Exception:  # pragma: no cover
        import traceback
        q.put("FAIL: " + traceback.format_exc())


@pytest.mark.timeout(120)
def test_single_rank_group_runs_every_collective_through_the_backend():
    """SIXDGS_DIST_SINGLE=1: a process group of ONE rank (round 4: how the RCCL path runs on the one-GPU box, tests/test_gpu_rccl_single.py);
    here over gloo: is_dist() turns true and the helpers take their collective branches."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_single_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=100)
    p.join(30)
    assert res == "ok", res
