"""Ray-sharded scoring (SURVEY 8(e) fallback): two ranks, each holding half of the rays, reproduce the single-GPU scorer
(global top-100 identical, scores within the rounding of the sum of exponentials).  The ranks share the one GPU of the
test box and talk over gloo; tools/ray_shard_check.py runs unchanged over RCCL on a multi-GPU node."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(400)
@pytest.mark.parametrize("mode,port", [("f16x3", 29551), ("f32", 29552)])
def test_two_ranks_ray_sharded_scorer_matches_single_gpu(mode, port):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    p = subprocess.run([sys.executable, "-W", "ignore", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), "tools/ray_shard_check.py", "--backend", "gloo",
                        "--device", "0", "--mode", mode, "--rays", "20037"],
                       cwd=ROOT, capture_output=True, text=True, timeout=380)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["ok"], d
    assert [r["rays"] for r in d["ranks"]] == [[0, 10112], [10112, 20037]]


@pytest.mark.timeout(400)
def test_two_ranks_ray_sharded_select_matches_single_gpu_select():
    """the select path over a ray-sharded scene (per-rank sweep, all-reduce of the sample statistics / g_t / key norm,
    all-gather of the per-rank top-k U -> global threshold, local candidates + local exact re-score, all-gather + merge): two ranks on
    one GPU over gloo return the single-GPU select answer -- the same 100 rays, values <= 2e-6 -- and, with a candidate budget that
    refuses every image, the ray-sharded two-pass scorer's (forced fall-back, all ranks deciding together)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    p = subprocess.run([sys.executable, "-W", "ignore", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29553", "tools/ray_shard_check.py", "--backend", "gloo",
                        "--device", "0", "--select", "--rays", "300037"],
                       cwd=ROOT, capture_output=True, text=True, timeout=380)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["ok"], d
    assert [r["rays"] for r in d["ranks"]] == [[0, 150272], [150272, 300037]]
    assert d["ranks"][0]["select"] == d["ranks"][1]["select"]                 # identical answers on every rank


@pytest.mark.timeout(500)
def test_bench_ray_parallelism_two_ranks_and_module_path():
    """`bench.py --parallelism ray`: every rank emits and keeps the key planes of its block of ellipsoids, all ranks score the same
    images, the selected rays are gathered from their owners for the pose solve -- two ranks on one GPU over gloo give the poses of
    the one-rank run of the same flag (which is the image-sharded run's scorer on the whole scene)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SIXDGS_BENCH_BACKEND="gloo", SIXDGS_BENCH_FORCE_DEVICE="0", SIXDGS_BENCH_DUMP_POSES="1")
    outs = []
    for n in (1, 2):
        p = subprocess.run([sys.executable, "-W", "ignore", "bench.py", "--gpus", str(n), "--parallelism", "ray", "--gaussians", "40000", "--batch", "3",
                            "--steps", "2", "--warmup", "1", "--skip-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=450)
        assert p.returncode == 0, p.stderr[-3000:]
        lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
        assert len(lines) == 1, p.stdout[-1000:]
        outs.append(json.loads(lines[0]))
    a, b = outs
    assert b["n_gpus"] == 2 and b["ranks_seen"] == 2 and b["scaling"] == "strong" and b["config"]["parallelism"].startswith("ray-sharded x2")
    assert a["config"]["rays"] == b["config"]["rays"] == 40000 * 64
    assert "select" in a["config"]["scoring_path"] and "select" in b["config"]["scoring_path"]
    import numpy as np
    pa, pb = np.asarray(a["poses_last_step"]), np.asarray(b["poses_last_step"])
    assert pa.shape == pb.shape == (3, 4, 4) and np.abs(pa - pb).max() < 1e-4
    assert abs(b["value"] - 3 * 2 / (b["ms_per_step"] * 2e-3)) < 1e-2 * b["value"]          # poses of ONE image set per step, not x world


def test_single_process_split_passes_equal_the_fused_scorer():
    """pass 1 + pass 2 with the statistics handed straight back are the fused scorer, bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import importlib
    import numpy as np
    sys.path.insert(0, ROOT)
    ops = importlib.import_module("6dgs_amd.ops")
    rng = np.random.default_rng(3)
    key = torch.from_numpy(rng.standard_normal((5000, 384)).astype(np.float32)).cuda()
    q = torch.from_numpy(rng.standard_normal((2, 256, 384)).astype(np.float32)).cuda()
    n_tok = torch.tensor([256, 99], dtype=torch.int32, device="cuda")
    q[1, 99:] = 0
    pl, sc = ops.split_planes_f16(key)
    idx0, val0, s0, st0 = ops.score_topk(q, n_tok, None, 100, want_stats=True, key_planes=pl, key_scale=sc)
    ws = torch.empty(ops.score_topk_workspace_bytes(5000, 2, 100), dtype=torch.uint8, device="cuda")
    st = ops.score_pass1(q, n_tok, None, ws, 100, key_planes=pl, key_scale=sc)
    assert torch.equal(st, st0)
    idx, val, s = ops.score_pass2(st, n_tok, 5000, ws, 100, used_planes=True)
    assert torch.equal(idx, idx0) and torch.equal(val, val0) and torch.equal(s, s0)
    small = torch.empty(ops.score_topk_workspace_bytes(5000, 1, 100), dtype=torch.uint8, device="cuda")
    with pytest.raises(RuntimeError):                      # both images must stay resident between the passes
        ops.score_pass1(q, n_tok, None, small, 100, key_planes=pl, key_scale=sc)


@pytest.mark.parametrize("mode", ["f16x3", "f32"])
def test_streamed_scorer_without_key_cache_matches_the_resident_one(mode):
    """score_tokens_streamed (keys recomputed per ray chunk, two sweeps, nothing of size R x 384 resident) returns the
    resident scorer's top-100."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import importlib
    import numpy as np
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module("6dgs_amd")
    ops = importlib.import_module("6dgs_amd.ops")
    syn = importlib.import_module("6dgs_amd.synthetic")
    ops.set_mma_mode({"f16x3": ops.MMA_F16X3, "f32": ops.MMA_F32}[mode])
    try:
        rays = syn.make_rays(9001, 2)
        ori, dr, rgb = (torch.from_numpy(rays[k]).cuda() for k in ("ori", "dir", "rgb"))
        idm = pkg.IdentificationModule("dino")
        idm.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_scorer_state_dict(0, with_cnn=True).items()}, strict=False)
        idm = idm.cuda().eval()
        toks = [torch.from_numpy(syn.make_tokens(t, 40 + i, 40.0)).cuda() for i, t in enumerate((256, 77, 1))]
        idx0, val0, _ = idm.score_tokens(toks, ori, dr, rgb, 100, want_scores=False)
        for chunk in (2048, 4096 + 128, 1 << 20):
            idx, val = idm.score_tokens_streamed(toks, ori, dr, rgb, 100, chunk_rays=chunk)
            assert torch.equal(idx, idx0), chunk
            assert float(((val - val0).abs() / val0.abs().clamp(min=1e-30)).max()) < 3e-6
    finally:
        ops.set_mma_mode(ops.MMA_DEFAULT)
