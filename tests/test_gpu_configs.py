"""BASELINE.json configs at their FULL workload on the GPU, each with a direct oracle check.

  headline  500 k Gaussians x 64 iso-cell rays = 32 M rays: one image's scores / top-100 against the CPU oracle
  cfg-2     ~300 k Gaussians written as a 3DGS point_cloud.ply, read back, B = 1, one 800x800 query, end to end
  cfg-3     1 M Gaussians x 64 = 64 M rays, 8 images on one rank (the scorer groups them: fewer logits blocks fit than images)
  cfg-4     2 M Gaussians x 256 = 512 M rays: (a) one rank's 64 M-ray share through the pass-1 / pass-2 cut that ray sharding
            and streaming use, against the resident scorer and the oracle; (b) the WHOLE 512 M rays through
            score_tokens_streamed with 16 images (the key planes, 786 GB, do not fit a GPU).

The oracle (oracle/sixdgs_oracle.c: three OpenMP sweeps of fp32 dot products, fp64 sums) scores ~0.75 M rays/s per image on
the 128-core host, so a full-size check costs 25-90 s of CPU per image.  It gets the GPU-produced fp32 keys (the ray MLP is
parity-tested against the oracle at small R in test_gpu_parity.py) and the boundary's own token block.

Tie policy at these sizes: with random weights the softmax over 3*10^7 rays is nearly flat and adjacent
top-101 scores differ by ~1e-7 relative, below fp32 resolution of the path.  Required: every ray the oracle ranks in the top
100 with a margin of more than 8e-6 * max score over the 101st is returned, and nothing is returned whose oracle score is
more than that margin below the oracle's 100th."""
import importlib
import os
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SCORE_TOL = 1e-5          # max |score - oracle| / max oracle score (the bar of the small-size parity tests)
MARGIN = 8e-6             # top-100 membership margin, relative to the largest score


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if torch.cuda.get_device_properties(0).total_memory < 200 * 2**30:
        pytest.skip("needs the 288 GB of an MI355X")
    pkg = importlib.import_module("6dgs_amd")
    syn = importlib.import_module("6dgs_amd.synthetic")
    ops = importlib.import_module("6dgs_amd.ops")
    ops.set_mma_mode(ops.MMA_DEFAULT)
    sd = syn.make_scorer_state_dict(0, with_cnn=True)
    idm = pkg.IdentificationModule("dino")
    idm.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    idm = idm.cuda().eval()
    yield dict(pkg=pkg, syn=syn, ops=ops, sd=sd, idm=idm)
    idm.invalidate_caches()
    torch.cuda.empty_cache()


def emit(env, n_gauss, k_rays, scene=None):
    pkg, syn = env["pkg"], env["syn"]
    if scene is None:
        scene = pkg.GaussianScene.from_dict(syn.make_scene(n_gauss, 0), device="cuda")
    ori, dr, rgb = pkg.generate_all_possible_rays(scene, max_ellipsoids=-1, emitter="isocell", rays_per_ellipsoid=k_rays)
    fin = torch.isfinite(dr).all(dim=1)
    if not bool(fin.all()):
        ori, dr, rgb = ori[fin].contiguous(), dr[fin].contiguous(), rgb[fin].contiguous()
    return scene, ori, dr, rgb


def host_keys(env, ori, dr, rgb, chunk=4_000_000):
    """fp32 keys K[R,384] of the HIP ray MLP + k_proj on the host (1536 B/ray), fetched chunk by chunk."""
    ops, w = env["ops"], env["idm"].packed_weights(ori.device)
    r = ori.shape[0]
    need = r * 384 * 4
    try:
        import psutil
        if psutil.virtual_memory().available < 1.3 * need:
            pytest.skip(f"host memory: {need / 2**30:.0f} GiB of fp32 keys for the oracle do not fit")
    except ImportError:
        pass
    out = np.empty((r, 384), np.float32)
    for r0 in range(0, r, chunk):
        r1 = min(r0 + chunk, r)
        _, key = ops.ray_keys(ori[r0:r1], dr[r0:r1], rgb[r0:r1], w)
        out[r0:r1] = key.cpu().numpy()
        del key
    return out


def sampled_key_parity(oracle, env, ori, dr, rgb, planes, inv, what, launch_rays=None, sample=None, n_stride=32768):
    """The ORACLE's ray MLP + k_proj (ray_preprocessor.py:11-46, our_multihead_attention.py:74 restated in C) on >= 32 k rays spread over the
    WHOLE scene against the HIP key planes the scorer actually streams (reconstructed (h + l) x the 128-ray tile's scale) -- VERDICT r3 #2:
    until round 4 the full-size configs fed the oracle GPU-produced keys, and the chain itself met the oracle at R <= 300 037 only.
    The sample: a uniform stride over all rays; the first / second / 128th / 129th and the last two rays of EVERY launch of the chain
    (`launch_rays` rays each: 125-2000 launches at these sizes); every ray of the ragged last tiles; and (sample=(indices, planes, inv)) a
    stride of the select path's ray sample, checked against ITS planes (a second pass of the chain over gathered rays)."""
    ops = env["ops"]
    R = int(ori.shape[0])
    L = int(launch_rays or ops.RAY_KEYS_CHUNK)
    picks = [np.linspace(0, R - 1, n_stride).astype(np.int64), np.arange(max(0, R - 300), R, dtype=np.int64)]
    starts = np.arange(0, R, L, dtype=np.int64)
    ends = np.minimum(starts + L, R)
    picks += [starts, np.minimum(starts + 1, R - 1), np.minimum(starts + 127, R - 1), np.minimum(starts + 128, R - 1), ends - 1, np.maximum(ends - 2, 0)]
    idx_np = np.unique(np.concatenate(picks))
    idx = torch.from_numpy(idx_np).cuda()

    def hip_rows(pl, iv, rows):
        p = pl[rows].contiguous().view(torch.float16).view(-1, 12, 2, 32).float()
        return ((p[:, :, 0] + p[:, :, 1]).reshape(-1, 384) * iv[rows // 128][:, None]).cpu().numpy()

    t0 = time.time()
    _, okey = oracle.ray_features(ori[idx].cpu().numpy(), dr[idx].cpu().numpy(), rgb[idx].cpu().numpy(), env["sd"], want_feat=False)
    dt = time.time() - t0
    hkey = hip_rows(planes, inv, idx)
    tile_max = np.abs(hkey).max()                                  # (the sampled rows' largest element: rows far below their tile's scale are held to it / 64)
    den = np.maximum(np.abs(okey).max(axis=1), tile_max / 64)
    err = np.abs(hkey - okey).max(axis=1) / den
    assert np.isfinite(hkey).all() and err.max() < 5e-6, f"{what}: key planes differ from the oracle's ray MLP + k_proj by {err.max():.2e} (row {idx_np[err.argmax()]})"
    msg = f"[{what}] key parity: {idx_np.shape[0]} rays of {R} ({starts.shape[0]} chain launches of {L}), oracle {dt:.1f} s, max row error {err.max():.2e}"
    if sample is not None:
        s_idx, s_pl, s_inv = sample
        sel = torch.arange(0, s_idx.shape[0], max(1, s_idx.shape[0] // 4096), device="cuda")
        rows = s_idx[sel]
        _, okey_s = oracle.ray_features(ori[rows].cpu().numpy(), dr[rows].cpu().numpy(), rgb[rows].cpu().numpy(), env["sd"], want_feat=False)
        hs = hip_rows(s_pl, s_inv, sel)
        es = np.abs(hs - okey_s).max(axis=1) / np.maximum(np.abs(okey_s).max(axis=1), np.abs(hs).max() / 64)
        assert es.max() < 5e-6, f"{what}: the select path's sample planes differ from the oracle by {es.max():.2e}"
        msg += f"; select sample: {sel.shape[0]} of {s_idx.shape[0]} rays, {es.max():.2e}"
    print(msg)


def oracle_check(oracle, env, key_np, tok_np, hip_scores, hip_idx, hip_val, what):
    """scores [R] and top-100 of ONE image against the oracle on the same keys and tokens."""
    t0 = time.time()
    s_ref = oracle.attention_scores(oracle.q_proj(tok_np, env["sd"]), key_np)
    dt = time.time() - t0
    smax = float(s_ref.max())
    if hip_scores is not None:
        err = float(np.abs(hip_scores.astype(np.float64) - s_ref).max() / smax)
        assert err < SCORE_TOL, f"{what}: scores differ from the oracle by {err:.2e} of the maximum"
    else:
        err = float("nan")
    order = np.argsort(-s_ref, kind="stable")
    s100, s101 = float(s_ref[order[99]]), float(s_ref[order[100]])
    margin = MARGIN * smax
    got = np.asarray(hip_idx, np.int64)
    assert len(set(got.tolist())) == 100 and got.min() >= 0 and got.max() < s_ref.shape[0]
    must = order[:100][s_ref[order[:100]] - s101 > margin]
    missing = set(must.tolist()) - set(got.tolist())
    assert not missing, f"{what}: {len(missing)} rays clearly inside the oracle's top-100 are missing"
    assert float(s_ref[got].min()) >= s100 - margin, f"{what}: a returned ray scores below the oracle's 100th by more than the margin"
    verr = float(np.abs(np.asarray(hip_val, np.float64) - s_ref[got]).max() / smax)
    assert verr < SCORE_TOL, f"{what}: top-100 values differ from the oracle's scores of those rays by {verr:.2e}"
    same = len(set(order[:100].tolist()) & set(got.tolist()))
    print(f"[{what}] R={s_ref.shape[0]} T={tok_np.shape[0]}: oracle {dt:.1f} s on {oracle.num_threads()} threads, score err {err:.2e}, "
          f"top-100 value err {verr:.2e}, {len(must)} rays with a clear margin all present, {same}/100 identical to the oracle's list")
    return s_ref


# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.timeout(600)
def test_headline_500k_x64_scores_and_top100_against_the_oracle(env, oracle):
    """The size bench.py's headline number is quoted on: 32 M rays, 256 tokens, one image against the oracle; a second image
    (173 tokens) in the same launch keeps the batched path honest."""
    ops, syn, idm = env["ops"], env["syn"], env["idm"]
    _, ori, dr, rgb = emit(env, 500_000, 64)
    assert ori.shape[0] > 31_000_000
    toks = [syn.make_tokens(t, 20 + i, 40.0) for i, t in enumerate((256, 173))]
    idx, val, sc = idm.score_tokens([torch.from_numpy(t).cuda() for t in toks], ori, dr, rgb, 100, want_scores=True)
    assert idm._key_cache["key"] is None and idm._key_cache["planes"].shape == (ori.shape[0], 1536)      # plane path, no fp32 keys resident
    assert idm.last_scoring_path == "two-pass"
    # the inference path proper (no score vector asked for): the select path -- no logits through HBM, candidates re-scored exactly
    i_s, v_s, none = idm.score_tokens([torch.from_numpy(t).cuda() for t in toks], ori, dr, rgb, 100, want_scores=False)
    assert none is None and idm.last_scoring_path == "select", idm.last_scoring_path
    print(f"[headline] select path: candidates examined per image {idm.last_select_candidates}")
    for b in range(2):
        assert set(i_s[b].tolist()) == set(idx[b].tolist())
        assert float((v_s[b] - sc[b][i_s[b]]).abs().max() / val[b][0]) < 2e-5
    kc = idm._key_cache
    sampled_key_parity(oracle, env, ori, dr, rgb, kc["planes"], kc["scale"], "headline 500k x 64",
                       sample=(ops.select_sample_indices(ori.shape[0], "cuda"), kc["sample"][0], kc["sample"][1]))
    key_np = host_keys(env, ori, dr, rgb)
    s_ref = oracle_check(oracle, env, key_np, toks[0], sc[0].cpu().numpy(), idx[0].cpu().numpy(), val[0].cpu().numpy(), "headline 500k x 64")
    # ... and the select path's answer against the same oracle scores
    got = i_s[0].cpu().numpy()
    order = np.argsort(-s_ref, kind="stable")
    margin = MARGIN * float(s_ref.max())
    must = order[:100][s_ref[order[:100]] - s_ref[order[100]] > margin]
    assert set(must.tolist()) <= set(got.tolist()) and float(s_ref[got].min()) >= float(s_ref[order[99]]) - margin
    assert np.abs(v_s[0].cpu().numpy() - s_ref[got]).max() / float(s_ref.max()) < SCORE_TOL
    print(f"[headline select] {len(set(order[:100].tolist()) & set(got.tolist()))}/100 identical to the oracle's list")
    del s_ref
    # pose tail at this size: the oracle's pose from the HIP top-100 (the same rays) within north_star's 1e-4
    up = torch.tensor([[0.2, -0.9, 0.3]], device="cuda")
    up = up / up.norm()
    sol = ops.solve_pose(ori, dr, idx[:1], val[:1], up)
    p_ref = oracle.pose_from_topk(ori.cpu().numpy(), dr.cpu().numpy(), idx[0].cpu().numpy(), val[0].cpu().numpy(), up[0].cpu().numpy())
    scale = max(1.0, float(np.abs(p_ref["c2w"]).max()))
    assert np.abs(sol["c2w"][0].cpu().numpy() - p_ref["c2w"]).max() / scale < 1e-4
    idm.invalidate_caches()


@pytest.mark.timeout(600)
def test_cfg2_300k_scene_through_a_ply_single_image_end_to_end(env, oracle, tmp_path):
    """configs[1]: a ~300 k-Gaussian pretrained scene arrives as point_cloud.ply; B = 1; one 800x800 query through the public
    callables (generate_all_possible_rays -> test_pose_estimation), then the same image's scores against the oracle."""
    pkg, syn, ops, idm = env["pkg"], env["syn"], env["ops"], env["idm"]
    src = pkg.GaussianScene.from_dict(syn.make_scene(300_000, 3), device="cuda")
    path = str(tmp_path / "point_cloud" / "iteration_30000" / "point_cloud.ply")
    src.save_ply(path)
    assert os.path.getsize(path) > 300_000 * 62 * 4
    scene = pkg.GaussianScene.load_ply(path, sh_degree=3, device="cuda")
    for f in ("_xyz", "_scaling", "_rotation", "_features_dc", "_features_rest", "_opacity"):
        assert torch.equal(getattr(scene, f), getattr(src, f)), f                      # the file round trip is bit-exact
    del src
    _, ori, dr, rgb = emit(env, 0, 64, scene=scene)
    assert 18_000_000 < ori.shape[0] <= 300_000 * 64
    cams = [pkg.CameraInfo(**c) for c in syn.make_cameras(1, 7, width=800, height=800)]
    res, t_err, a_err, _, _ = pkg.test_pose_estimation(cams, idm, ori, dr, rgb, torch.tensor([0.0, 1.0, 0.0]), sequence_id="ignatius",
                                                       category_id="tandt", verbose=False)
    assert len(res) == 1 and np.isfinite(t_err) and np.isfinite(a_err)
    c2w = np.asarray(res[0]["pred_c2w"], np.float64)
    assert c2w.shape == (4, 4) and np.allclose(c2w[3], [0, 0, 0, 1]) and abs(np.linalg.det(c2w[:3, :3]) - 1.0) < 1e-3
    # the same image, stage by stage: tokens from the module's own image side, scores / top-100 / pose against the oracle
    tp = importlib.import_module("6dgs_amd.test")
    img, mask = tp.prepare_image(cams[0].image, "cuda")
    toks, fmaps = idm.image_tokens([img], [None])
    up = idm.camera_up(fmaps)
    tl = list(toks) if torch.is_tensor(toks) else toks
    idx, val, sc = idm.score_tokens(tl, ori, dr, rgb, 100, want_scores=True)
    key_np = host_keys(env, ori, dr, rgb)
    oracle_check(oracle, env, key_np, tl[0].cpu().numpy(), sc[0].cpu().numpy(), idx[0].cpu().numpy(), val[0].cpu().numpy(), "cfg-2 300k PLY")
    p_ref = oracle.pose_from_topk(ori.cpu().numpy(), dr.cpu().numpy(), idx[0].cpu().numpy(), val[0].cpu().numpy(), up[0].cpu().numpy())
    scale = max(1.0, float(np.abs(p_ref["c2w"]).max()))
    assert np.abs(c2w - p_ref["c2w"]).max() / scale < 1e-4                             # the pose test_pose_estimation returned
    idm.invalidate_caches()


@pytest.mark.timeout(900)
def test_cfg3_1m_x64_eight_images_per_rank_grouped(env, oracle):
    """configs[2]: 64 views over 8 GPUs = 8 images per rank against 64 M rays (98 GB of key planes).  The logits of 8 images
    (8 x 50 GB) do not fit next to them, so the library scores the batch in groups -- the grouping has to be invisible."""
    ops, syn, idm = env["ops"], env["syn"], env["idm"]
    _, ori, dr, rgb = emit(env, 1_000_000, 64)
    R = ori.shape[0]
    assert R > 63_000_000
    n_t = (256, 256, 137, 256, 200, 256, 1, 256)
    toks = [torch.from_numpy(syn.make_tokens(t, 40 + i, 40.0)).cuda() for i, t in enumerate(n_t)]
    kc = idm._ensure_keys(ori, dr, rgb)
    # the 8 images through the select path in ONE launch (20 B of workspace per ray and image instead of 784: no grouping needed)
    i_s, v_s, _ = idm.score_tokens(toks, ori, dr, rgb, 100, want_scores=False)
    assert idm.last_scoring_path == "select", idm.last_scoring_path
    print(f"[cfg-3] select path: candidates examined per image {idm.last_select_candidates}")
    idm._select_ws = None                                                                                            # 10 GB back before the 150 GB below
    ws = torch.empty(ops.score_topk_workspace_bytes(R, 3, 100, planes=True), dtype=torch.uint8, device="cuda")      # 3 of 8 images fit
    idx, val, sc = idm.score_tokens(toks, ori, dr, rgb, 100, want_scores=True, workspace=ws)
    assert idx.shape == (8, 100) and sc.shape == (8, R)
    tot = sc.double().sum(dim=1).cpu().numpy()
    assert np.allclose(tot, n_t, rtol=2e-4)                                            # softmax mass per image = its token count
    for b in (2, 7):        # an image of the first group and the lone image of the last one: alone == inside the grouped batch
        i1, v1, s1 = idm.score_tokens([toks[b]], ori, dr, rgb, 100, want_scores=True, workspace=ws)
        assert torch.equal(i1[0], idx[b]) and torch.equal(v1[0], val[b]) and torch.equal(s1[0], sc[b])
        del s1
    for b8 in range(8):
        if n_t[b8] > 1:
            assert set(i_s[b8].tolist()) == set(idx[b8].tolist()), b8
            assert float((v_s[b8] - sc[b8][i_s[b8]]).abs().max() / val[b8][0]) < 2e-5
    b = 2                   # 137 tokens: ragged token count + about half the oracle time of a full image
    s_b, i_b, v_b = sc[b].cpu().numpy(), idx[b].cpu().numpy(), val[b].cpu().numpy()
    del sc, ws
    sampled_key_parity(oracle, env, ori, dr, rgb, kc["planes"], kc["scale"], "cfg-3 1M x 64",
                       sample=(ops.select_sample_indices(R, "cuda"), kc["sample"][0], kc["sample"][1]))
    key_np = host_keys(env, ori, dr, rgb)
    oracle_check(oracle, env, key_np, toks[b].cpu().numpy(), s_b, i_b, v_b, "cfg-3 1M x 64, image 2 of 8")
    assert kc["planes"].shape[0] == R
    idm.invalidate_caches()


@pytest.mark.timeout(1200)
def test_cfg4_2m_x256_rank_share_and_whole_scene_streamed(env, oracle):
    """configs[3]: 2 M Gaussians x 256 rays = 512 M rays, 16 images per rank."""
    ops, syn, idm = env["ops"], env["syn"], env["idm"]
    _, ori, dr, rgb = emit(env, 2_000_000, 256)
    R = ori.shape[0]
    assert R > 500_000_000
    n_t = (256, 137, 256, 256, 200, 256, 256, 256, 64, 256, 256, 256, 256, 173, 256, 256)
    toks = [torch.from_numpy(syn.make_tokens(t, 60 + i, 40.0)).cuda() for i, t in enumerate(n_t)]
    # (a) one rank's share under ray sharding over 8 GPUs: the first 64 M rays.  The pass-1 / merge / pass-2 cut (what a ray
    # shard runs, and what streaming runs per chunk) against the resident scorer on the same rays, then against the oracle.
    S = 64_000_000
    o_s, d_s, c_s = ori[:S], dr[:S], rgb[:S]
    i_st, v_st, glob, mass = idm.score_tokens_streamed(toks[:2], o_s, d_s, c_s, 100, chunk_rays=8_388_608, return_stats=True)
    assert np.allclose(mass.cpu().numpy(), n_t[:2], rtol=2e-4)
    # the streamed call kept its chunks' key planes between the sweeps: eight 13 GB blocks now sit in PyTorch's caching allocator, which
    # cannot merge them into the 98 GB of planes + 94 GB of logits the resident call needs next (seen once as an out-of-memory error with
    # 124 GB "reserved but unallocated")
    torch.cuda.empty_cache()
    i_rs, v_rs, sc = idm.score_tokens(toks[:2], o_s, d_s, c_s, 100, want_scores=True)
    for b in range(2):
        assert set(i_st[b].tolist()) == set(i_rs[b].tolist())                                          # same 100 rays ...
        assert float((v_st[b] - v_rs[b]).abs().max() / v_rs[b].max()) < 2e-6                            # ... same values up to the rounding of sum-exp
    s_1, i_1, v_1 = sc[1].cpu().numpy(), i_st[1].cpu().numpy(), v_st[1].cpu().numpy()
    del sc
    kc = idm._key_cache
    sampled_key_parity(oracle, env, o_s, d_s, c_s, kc["planes"], kc["scale"], "cfg-4 rank share 64M rays (256 rays per ellipsoid)")
    # ... and the chunks of the WHOLE scene's streamed sweep: the chain over three of its 64 chunks (first, one in the middle, the ragged last), each
    # checked over its own launches (what score_tokens_streamed computes, uses and drops)
    wts = idm.packed_weights("cuda")
    for c0 in (0, 31 * 8_388_608, (R // 8_388_608) * 8_388_608):
        c1 = min(c0 + 8_388_608, R)
        if c1 <= c0:
            continue
        _, _, (pl_c, inv_c) = ops.ray_keys(ori[c0:c1], dr[c0:c1], rgb[c0:c1], wts, want_key=False, want_planes=True)
        sampled_key_parity(oracle, env, ori[c0:c1], dr[c0:c1], rgb[c0:c1], pl_c, inv_c, f"cfg-4 streamed chunk at ray {c0}", n_stride=4096)
        del pl_c, inv_c
    idm.invalidate_caches()
    key_np = host_keys(env, o_s, d_s, c_s)
    s_ref = oracle_check(oracle, env, key_np, toks[1].cpu().numpy(), s_1, i_1, v_1, "cfg-4 rank share 64M rays (streamed cut)")
    del key_np, s_ref
    torch.cuda.empty_cache()
    # (b) the whole scene, 16 images, streamed: 2 sweeps over 64 chunks of 8 M rays
    t0 = time.time()
    idx, val, glob, mass = idm.score_tokens_streamed(toks, ori, dr, rgb, 100, chunk_rays=8_388_608, return_stats=True)
    torch.cuda.synchronize()
    dt = time.time() - t0
    assert idx.shape == (16, 100) and np.allclose(mass.cpu().numpy(), n_t, rtol=3e-4)       # softmax mass over all 512 M rays
    for b in range(16):
        v, i = val[b], idx[b]
        assert bool((v[:-1] >= v[1:]).all()) and len(set(i.tolist())) == 100 and int(i.min()) >= 0 and int(i.max()) < R
    # the returned values are the scores of the returned rays: recompute them on the host in fp64 from the keys of just those
    # rays and the global row statistics of sweep 1
    w = idm.packed_weights("cuda")
    for b in (0, 8):
        sel = idx[b]
        _, key = ops.ray_keys(ori[sel].contiguous(), dr[sel].contiguous(), rgb[sel].contiguous(), w)
        q = oracle.q_proj(toks[b].cpu().numpy(), env["sd"]).astype(np.float64)
        logit = q @ key.cpu().numpy().astype(np.float64).T / np.sqrt(384.0)                                # [T, 100]
        st = glob[b, : n_t[b]].cpu().numpy().astype(np.float64)
        ref = (np.exp(logit - st[:, :1]) / st[:, 1:2]).sum(axis=0)
        assert np.abs(val[b].cpu().numpy() - ref).max() / ref.max() < 1e-5
    print(f"[cfg-4 whole scene] R={R} rays x 16 images streamed in {dt:.1f} s ({16 / dt:.2f} poses/s on one GPU)")
    del glob, mass
    # (c) the same through the streamed SELECT path: one sweep (+ the 1/16 sample) instead of two, U = 4 B per ray and image
    idm._stream_sample = None
    t0 = time.time()
    i_s, v_s = idm.score_tokens_streamed(toks, ori, dr, rgb, 100, chunk_rays=8_388_608)
    torch.cuda.synchronize()
    dt_s = time.time() - t0
    assert idm.last_scoring_path == "streamed select", idm.last_scoring_path
    for b in range(16):
        if n_t[b] > 1:
            assert set(i_s[b].tolist()) == set(idx[b].tolist()), b
            assert float((v_s[b] - val[b]).abs().max() / val[b][0]) < 2e-5
    print(f"[cfg-4 whole scene, select] {dt_s:.1f} s incl. the sample's keys ({16 / dt_s:.2f} poses/s); candidates {idm.last_select_candidates}")
