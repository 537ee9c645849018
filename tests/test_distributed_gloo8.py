"""EIGHT-rank rehearsal of the multi-GPU plumbing over gloo on CPU tensors: the first 8-GPU run is the driver's,
unattended, and nothing beyond two ranks had ever executed.  Uneven and EMPTY shards (13 and 5 test views over 8 ranks), one failing
rank among eight, a rank 0 that spends longer in a long-wait stage than the default group's timeout, and the ray-sharded
merges with a shard that holds no ray at all."""
import importlib
import os
import socket
import sys
import time

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD = 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(rank, world, port, **env):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **env)
    torch.set_num_threads(1)
    dd = importlib.import_module("6dgs_amd.distributed")
    r, w, _ = dd.init_from_env("gloo")
    assert (r, w) == (rank, world) and dd.is_dist() and dd.world() == world
    return dd


def _run(target, world=WORLD, timeout=240, **kw):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_entry, args=(target, r, world, port, q), kwargs=kw) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=timeout) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(res) == [(r, "ok") for r in range(world)], [x for x in res if x[1] != "ok"]


def _entry(name, rank, world, port, q, **kw):
    try:
        globals()[name](rank, world, port, **kw)
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))
        raise


def _image_worker(rank, world, port):
    dd = _setup(rank, world, port)
    pkg = importlib.import_module("6dgs_amd")
    syn = importlib.import_module("6dgs_amd.synthetic")
    assert dd.ranks_seen("cpu") == world
    # scene + weights from rank 0 to seven peers
    ref = syn.make_scene(301, 3)
    scene = dd.broadcast_scene(pkg.GaussianScene.from_dict(ref, device="cpu") if rank == 0 else None, 0, device="cpu")
    assert len(scene) == 301 and np.array_equal(scene._xyz.numpy(), ref["xyz"]) and np.array_equal(scene._features_rest.numpy(), ref["f_rest"])
    lin = torch.nn.Linear(7, 3)
    with torch.no_grad():
        lin.weight.fill_(float(rank + 1))
        lin.bias.fill_(float(-rank))
    dd.broadcast_module(lin, 0)
    assert float(lin.weight[2, 6]) == 1.0 and float(lin.bias[1]) == 0.0
    # 13 views over 8 ranks: blocks of 2,2,2,2,2,1,1,1;  5 views: 1,1,1,1,1,0,0,0 (three ranks hold NO image);  0 views: nobody does
    for n_img, want in ((13, [2, 2, 2, 2, 2, 1, 1, 1]), (5, [1, 1, 1, 1, 1, 0, 0, 0]), (0, [0] * 8), (8, [1] * 8)):
        counts = dd.shard_counts(n_img, world)
        assert counts == want and sum(counts) == n_img
        lo, hi = dd.shard_range(n_img, rank, world)
        assert hi - lo == counts[rank] and lo == sum(counts[:rank])
        c2w = torch.stack([torch.eye(4) * (i + 1) for i in range(lo, hi)]) if hi > lo else torch.zeros(0, 4, 4)
        status = torch.arange(lo, hi, dtype=torch.int32)
        for kw in ({}, {"counts": counts}):                # sizes exchanged first / ONE fixed-size gather
            allp, alls = dd.gather_poses(c2w, status, 0, **kw)
            if rank == 0:
                assert allp.shape == (n_img, 4, 4) and alls.tolist() == list(range(n_img))
                assert [float(allp[i, 0, 0]) for i in range(n_img)] == [float(i + 1) for i in range(n_img)]
            else:
                assert allp is None and alls is None
        res = dd.gather_results([{"frame_id": i, "rank": rank} for i in range(lo, hi)], 0)
        if rank == 0:
            assert [r["frame_id"] for r in res] == list(range(n_img))
            assert [r["rank"] for r in res] == [r_ for r_ in range(world) for _ in range(counts[r_])]
        else:
            assert len(res) == hi - lo
        assert dd.all_counts(hi - lo, "cpu") == counts
    # counts that do not describe this rank's block are refused before any collective (every rank raises: nobody is left waiting)
    with pytest.raises(RuntimeError, match="do not describe"):
        dd.gather_poses(torch.zeros(3, 4, 4), None, 0, counts=[1] * world)
    assert dd.max_over_ranks(float(rank), "cpu") == float(world - 1)
    assert dd.broadcast_int(1234 + rank, 0, "cpu") == 1234
    # one failing rank among eight: all eight leave the stage, the next collective pairs up
    def stage():
        if rank == 5:
            raise KeyError("model_state_dict")
        return rank

    try:
        dd.agree(stage, "load scene")
        raised = None
    except (RuntimeError, KeyError) as e:
        raised = e
    assert isinstance(raised, KeyError if rank == 5 else RuntimeError), raised
    if rank != 5:
        assert "another rank failed during 'load scene'" in str(raised)
    t = torch.tensor([float(rank + 1)])
    torch.distributed.broadcast(t, 0)
    assert float(t) == 1.0
    assert dd.agree(lambda: rank, "fine", long_wait=True) == rank
    dd.barrier()


def _slow_rank0_worker(rank, world, port, sleep_s=9.0):
    """rank 0 spends longer inside a long-wait stage (training) than the DEFAULT group's timeout; the peers wait for it in
    the long-wait group's all-reduce -- which exists, communicator included, since init_from_env -- and do not time out."""
    dd = _setup(rank, world, port, SIXDGS_DIST_TIMEOUT_S="4", SIXDGS_DIST_LONG_TIMEOUT_S="120")
    assert dd._long_group is not None and dd._long_group_timeout_s == 120          # created and warmed by init_from_env
    g = dd.long_wait_group()
    assert g is dd._long_group

    def stage_train():
        if rank == 0:
            time.sleep(sleep_s)
        return "trained" if rank == 0 else "waited"

    t0 = time.time()
    out = dd.agree(stage_train, "train the scorer (rank 0)", "cpu", long_wait=True)
    waited = time.time() - t0
    assert out == ("trained" if rank == 0 else "waited")
    assert waited >= sleep_s - 0.5, waited                                           # the peers did wait for rank 0 ...
    t = torch.tensor([float(rank)])
    torch.distributed.all_reduce(t)                                                  # ... and the default group still works afterwards
    assert float(t) == sum(range(world))
    dd.barrier()


def _ray_shard_worker(rank, world, port):
    dd = _setup(rank, world, port)
    rng = np.random.default_rng(11)
    B, T, K = 3, 256, 100
    # ray slices of VERY different sizes, one of them EMPTY (rank 6) and one shorter than k (rank 7: 37 rays)
    sizes = [400, 250, 1, 300, 129, 128, 0, 37]
    assert len(sizes) == world
    R = sum(sizes)
    lo = sum(sizes[:rank])
    hi = lo + sizes[rank]
    assert dd.all_counts(hi - lo, "cpu") == sizes
    logits = (rng.standard_normal((B, T, R)) * 6).astype(np.float32)
    logits[1, 40:] = -np.inf
    loc = logits[:, :, lo:hi].astype(np.float64)
    if hi > lo:
        m = loc.max(axis=2)
        with np.errstate(invalid="ignore"):
            sl = np.where(np.isinf(m), 0.0, np.exp(loc - np.where(np.isinf(m), 0.0, m)[..., None]).sum(axis=2))
    else:
        m, sl = np.full((B, T), -np.inf), np.zeros((B, T))           # what sixdgs_score_pass1 leaves for a slice without rays
    g = dd.merge_row_stats(torch.from_numpy(np.stack([m, sl], axis=-1).astype(np.float32))).numpy().astype(np.float64)
    M = logits.astype(np.float64).max(axis=2)
    with np.errstate(invalid="ignore"):
        S = np.where(np.isinf(M), 0.0, np.exp(logits.astype(np.float64) - np.where(np.isinf(M), 0.0, M)[..., None]).sum(axis=2))
    assert np.array_equal(g[..., 0], M.astype(np.float32).astype(np.float64))
    assert np.allclose(g[..., 1], S, rtol=2e-6, atol=0)
    # candidate merge: per-rank top-k lists padded with (-1, NaN) where the slice holds fewer than k rays (or none)
    scores = rng.integers(0, 60, size=(B, R)).astype(np.float32)       # heavy ties across ranks: the lower global index wins
    sc_loc = scores[:, lo:hi]
    n_loc = hi - lo
    idx = torch.full((B, K), -1, dtype=torch.int64)
    val = torch.full((B, K), float("nan"))
    if n_loc:
        order = np.lexsort((np.broadcast_to(np.arange(n_loc), sc_loc.shape), -sc_loc), axis=1)[:, :K]
        idx[:, : order.shape[1]] = torch.from_numpy(order.astype(np.int64))
        val[:, : order.shape[1]] = torch.from_numpy(np.take_along_axis(sc_loc, order, axis=1))
    gi, gv = dd.merge_topk(idx, val, lo, K)
    ro = np.lexsort((np.broadcast_to(np.arange(R), scores.shape), -scores), axis=1)[:, :K]
    assert np.array_equal(gi.numpy(), ro) and np.array_equal(gv.numpy(), np.take_along_axis(scores, ro, axis=1))
    # U_(k) of the scene from the shards' lists
    u = rng.random((B, R)).astype(np.float32)
    ul = np.full((B, K), np.nan, np.float32)
    if n_loc:
        top = np.sort(u[:, lo:hi], axis=1)[:, ::-1][:, :K]
        ul[:, : top.shape[1]] = top
    uk = dd.kth_largest_of_union(torch.from_numpy(ul), K).numpy()
    assert np.array_equal(uk, np.sort(u, axis=1)[:, ::-1][:, K - 1])
    # the selected rays put together from their owners (the empty shard owns none)
    ori = torch.from_numpy(rng.standard_normal((R, 3)).astype(np.float32))
    dr = torch.from_numpy(rng.standard_normal((R, 3)).astype(np.float32))
    gsel = torch.from_numpy(rng.integers(0, R, size=(B, 23)))
    gsel[2, 5] = -1
    so, sd_ = dd.gather_selected_rays(gsel, ori[lo:hi], dr[lo:hi], lo)
    want_o, want_d = ori[gsel.clamp(min=0)], dr[gsel.clamp(min=0)]
    want_o[2, 5], want_d[2, 5] = 0.0, 0.0
    assert torch.equal(so, want_o) and torch.equal(sd_, want_d)
    dd.barrier()


@pytest.mark.timeout(300)
def test_eight_ranks_image_sharding_uneven_and_empty_blocks():
    _run("_image_worker")


@pytest.mark.timeout(300)
def test_eight_ranks_ray_sharded_merges_with_an_empty_shard():
    _run("_ray_shard_worker")


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 8])
def test_slow_rank0_in_a_long_wait_stage_outlasts_the_default_timeout(world):
    _run("_slow_rank0_worker", world=world)
