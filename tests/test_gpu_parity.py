"""GPU parity tests: every entry point of the C ABI (through 6dgs_amd.ops) against the CPU oracle and the
reference-generated golden vectors, on the same seeded inputs.  Run with `-m gpu` on an MI355X.

Tolerances are stated per test.  Integer / index / boolean results must be identical except at the
documented floating-point tie boundaries (arc-length table look-ups within one ulp of an entry).
"""
import importlib
import os

import numpy as np
import pytest

from conftest import quadricell_tie_cells, rel_err

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return importlib.import_module("6dgs_amd.ops")


def G(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def N(t):
    return t.detach().cpu().numpy()


def test_library_is_the_hip_extension(ops):
    lib = importlib.import_module("6dgs_amd._lib")
    assert lib.load().sixdgs_abi_version() == lib.ABI_VERSION >= 3
    with open("/proc/self/maps") as f:
        assert "lib6dgs_hip.so" in f.read()


def test_a2_mask_degraded(ops, oracle, golden):
    g = golden("g1_quadricell")
    m = N(ops.mask_degraded(G(np.log(g["a2_scale"]))))
    # exp(log(s)) differs from s by an ulp or two; compare against the oracle on the SAME activated values
    act = np.exp(np.log(g["a2_scale"]).astype(np.float32)).astype(np.float32)
    ref = oracle.mask_degraded(act)
    assert (m != ref).sum() <= 2
    assert (m != g["a2_mask"]).sum() <= 4


def test_a5_sym_eig(ops, oracle, golden):
    g = golden("g2_normals")
    vals, vecs = ops.sym_eig_3x3(G(g["mats"]))
    ov, oe = oracle.sym_eig_3x3(g["mats"])
    nspd = 200
    scale = np.abs(ov[:nspd]).max(1, keepdims=True)
    assert (np.abs(N(vals)[:nspd] - ov[:nspd]) / scale).max() < 2e-5
    gap = np.minimum(ov[:nspd, 1] - ov[:nspd, 0], ov[:nspd, 2] - ov[:nspd, 1]) / scale[:, 0]
    ok = gap > 1e-2
    assert np.abs(N(vecs)[:nspd][ok] - oe[:nspd][ok]).max() < 2e-3
    assert np.abs(N(vecs)[:nspd][ok] - g["eigvecs"][:nspd][ok]).max() < 2e-3


def test_a4_normals_knn(ops, oracle, golden):
    g = golden("g2_normals")
    n, knn = ops.normals_knn(G(g["pts"]), G(g["pts"]), 20, return_knn=True)
    on, oknn = oracle.compute_normals(g["pts"], g["pts"], 20, return_knn=True)
    assert (N(knn) == oknn).all()                     # same distances, same tie rule -> identical lists
    assert np.abs(N(n) - on).max() < 2e-4
    same = np.array([set(a) == set(b) for a, b in zip(N(knn), g["knn"])])
    assert same.mean() > 0.99
    assert np.abs(N(n)[same] - g["normals"][same]).max() < 5e-4


@pytest.mark.parametrize("kind", ["gauss", "clustered", "duplicates", "plane", "tiny"])
def test_a4_grid_knn_is_the_brute_force_search(ops, kind):
    """The uniform-grid search returns the brute-force neighbour lists (order included) and normals bit for bit."""
    rng = np.random.default_rng(4)
    n = 60_000
    if kind == "gauss":
        c = rng.standard_normal((n, 3))
    elif kind == "clustered":                         # dense blobs + far outliers: many empty cells, deep shells
        c = np.concatenate([rng.standard_normal((n - 200, 3)) * 0.01 + rng.integers(0, 3, (n - 200, 1)), rng.standard_normal((200, 3)) * 50])
    elif kind == "duplicates":                        # exact distance ties -> lowest index first
        c = np.repeat(rng.standard_normal((n // 4, 3)), 4, axis=0)
    elif kind == "plane":                             # degenerate extent along one axis
        c = rng.standard_normal((n, 3)) * np.array([1.0, 1.0, 0.0])
    else:
        n = 30
        c = rng.standard_normal((n, 3))
    c = G(c.astype(np.float32))
    nb, kb = ops.normals_knn(c, c, 20, return_knn=True, method="brute")
    ng, kg = ops.normals_knn(c, c, 20, return_knn=True, method="grid")
    assert torch.equal(kb, kg)
    assert torch.equal(torch.nan_to_num(nb), torch.nan_to_num(ng)) and torch.equal(torch.isnan(nb), torch.isnan(ng))
    q = G((rng.standard_normal((777, 3)) * 1.5).astype(np.float32))       # queries that are not cloud points, some outside the box
    nb, kb = ops.normals_knn(q, c, 20, return_knn=True, method="brute")
    ng, kg = ops.normals_knn(q, c, 20, return_knn=True, method="grid")
    assert torch.equal(kb, kg) and torch.equal(torch.nan_to_num(nb), torch.nan_to_num(ng))


def test_a4_normals_knn_chunked_queries(ops, oracle):
    rng = np.random.default_rng(3)
    cloud = rng.standard_normal((2500, 3)).astype(np.float32)
    q = cloud[100:900]
    n = ops.normals_knn(G(q), G(cloud), 20)
    assert np.abs(N(n) - oracle.compute_normals(q, cloud, 20)).max() < 5e-4


@pytest.mark.parametrize("P", [50, 64, 256])
def test_a6_quadricell_centers(ops, oracle, golden, P):
    g = golden("g1_quadricell")
    pts, eid = ops.quadricell_centers(G(g["scale"]), P)
    ref = g[f"P{P}_points"]
    assert tuple(pts.shape) == ref.shape
    assert (N(eid) == g[f"P{P}_eid"]).all()
    d = np.abs(N(pts) - ref).max(1)
    assert np.abs(N(pts)[:, 2] - ref[:, 2]).max() < 1e-7
    ties = quadricell_tie_cells(g[f"P{P}_eid"], ref)       # exact ties of the reference's table: see conftest
    assert (d[~ties] < 1e-6).all(), int((d[~ties] >= 1e-6).sum())
    smax = g["scale"][g[f"P{P}_eid"]].max(1)
    assert (d[ties] <= 2.5 * (2 * np.pi / 999) * smax[ties]).all()


@pytest.mark.parametrize("P", [50, 64, 256])
def test_a1_a6_a7_a10_emit_quadricell(ops, oracle, golden, syn, P):
    """Full emitter on the g1 ellipsoids (activated scales passed directly): ray set identical to the
    reference's, values within 1e-6 except the documented table ties."""
    g = golden("g1_quadricell")
    E = g["scale"].shape[0]
    rng = np.random.default_rng(9)
    f_dc = (0.3 * rng.standard_normal((E, 1, 3))).astype(np.float32)
    f_rest = (0.3 * rng.standard_normal((E, 15, 3))).astype(np.float32)
    ori, dr, rgb, src, n_cells = ops.emit_quadricell(G(g["xyz"]), G(g["scale"]), G(g["rot"]), G(f_dc), G(f_rest), 3, None,
                                                     G(g["normals"]), P, scale_is_log=False)
    assert n_cells == g[f"P{P}_points"].shape[0]
    ref_ori, ref_dir, ref_mid = g[f"P{P}_ori"], g[f"P{P}_dir"], g[f"P{P}_mid"]
    assert ori.shape[0] == ref_ori.shape[0]
    assert (N(src) == ref_mid).all()
    d = np.maximum(np.abs(N(ori) - ref_ori).max(1), np.abs(N(dr) - ref_dir).max(1) * g["scale"][ref_mid].max(1))
    # map the tie mask of the cells onto the rays that survived the hemisphere mask (order preserved)
    Rm = g["rotmat"][g[f"P{P}_eid"]]
    pw = np.einsum("nij,nj->ni", Rm, g[f"P{P}_points"])
    kept = g["normals"][g[f"P{P}_eid"], 0] * pw[:, 0] > 0
    assert kept.sum() == ref_ori.shape[0]
    ties = quadricell_tie_cells(g[f"P{P}_eid"], g[f"P{P}_points"])[kept]
    assert (d[~ties] < 2e-6).all(), int((d[~ties] >= 2e-6).sum())
    # colour: oracle SH on the emitted directions with the per-ray coefficients
    sh = np.concatenate([f_dc, f_rest], 1).transpose(0, 2, 1)[N(src)]          # [R,3,16]
    assert np.abs(N(rgb) - oracle.eval_sh_color(sh, N(dr), 3)).max() < 2e-6


def test_emit_quadricell_subset_and_log_scale(ops, oracle, golden):
    g = golden("g1_quadricell")
    E = g["scale"].shape[0]
    sel = np.array([5, 3, 40, 17, 17, 0], np.int64)
    nrm = g["normals"][: len(sel)]
    logs = np.log(g["scale"]).astype(np.float32)
    ori, dr, rgb, src, _ = ops.emit_quadricell(G(g["xyz"]), G(logs), G(g["rot"]), None, None, 0, G(sel), G(nrm), 50,
                                               scale_is_log=True, want_rgb=False)
    act = np.exp(logs).astype(np.float32)
    pts, eid = oracle.quadricell_centers(act[sel], 50)
    o_ori, o_dir, o_mid = oracle.mask_and_compute_rays(pts, eid, nrm, g["xyz"][sel], oracle.build_rotation(g["rot"][sel]))
    assert ori.shape[0] == o_ori.shape[0]
    assert (N(src) == sel[o_mid]).all()
    d = np.abs(N(ori) - o_ori).max(1)
    Rm = oracle.build_rotation(g["rot"][sel])[eid]
    kept = nrm[eid, 0] * np.einsum("nij,nj->ni", Rm, pts)[:, 0] > 0
    ties = quadricell_tie_cells(eid, pts)[kept]            # only the structural table ties may differ
    assert (d[~ties] < 2e-6).all()


@pytest.mark.parametrize("tgt,n0", [(35, 3), (64, 1), (256, 1), (50, 1)])
def test_a8_isocell_distribution(ops, golden, tgt, n0):
    g = golden("g3_isocell")
    d = N(ops.isocell_distribution(tgt, n0))
    assert d.shape == g[f"dirs_{tgt}_{n0}"].shape
    assert np.abs(d - g[f"dirs_{tgt}_{n0}"]).max() < 5e-7


@pytest.mark.parametrize("tgt", [64, 256])
def test_a9_rotate_isocell(ops, golden, tgt):
    g = golden("g3_isocell")
    r = N(ops.rotate_isocell(G(g[f"dirs_{tgt}_1"]), G(g["normals"])))
    ref = g[f"rot_{tgt}"]
    assert (np.isnan(r) == np.isnan(ref)).all()
    assert np.nanmax(np.abs(r - ref)) < 1e-6


def test_emit_isocell(ops, oracle, syn):
    sc = syn.make_scene(500, 1)
    rng = np.random.default_rng(2)
    nrm = rng.standard_normal((500, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    dirs = ops.isocell_distribution(64, 1)
    ori, dr, rgb, src = ops.emit_isocell(G(sc["xyz"]), G(sc["log_scale"]), G(sc["rot"]), G(sc["f_dc"]), G(sc["f_rest"]), 3, None,
                                         G(nrm), dirs)
    K = dirs.shape[0]
    assert ori.shape[0] == 500 * K
    ref_dir = oracle.rotate_isocell(N(dirs), nrm).reshape(-1, 3)
    assert np.abs(N(dr) - ref_dir).max() < 1e-6
    # origins lie on the ellipsoid surface along the ray direction from the centre
    R = oracle.build_rotation(sc["rot"])
    s = np.exp(sc["log_scale"])
    off = (N(ori).reshape(500, K, 3) - sc["xyz"][:, None]).astype(np.float64)
    loc = np.einsum("eji,ekj->eki", R.astype(np.float64), off)
    assert np.abs(((loc / s[:, None]) ** 2).sum(-1) - 1).max() < 1e-4
    cosang = (off * N(dr).reshape(500, K, 3)).sum(-1) / np.linalg.norm(off, axis=-1)
    assert cosang.min() > 1 - 1e-5
    sh = np.concatenate([sc["f_dc"], sc["f_rest"]], 1).transpose(0, 2, 1)[N(src)]
    assert np.abs(N(rgb) - oracle.eval_sh_color(sh, N(dr), 3)).max() < 2e-6
    assert (N(src).reshape(500, K) == np.arange(500)[:, None]).all()


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_a10_sh_colour(ops, golden, deg):
    g = golden("g4_sh")
    c = N(ops.eval_sh_color(G(g["sh"]), G(g["dir"]), deg))
    assert np.abs(c - g[f"rgb_deg{deg}"]).max() < 1e-6


# ------------------------------------------------------------------------------------------------
# scorer
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module", params=["bf16x6", "f16x3", "f16x3l32", "f32"])
def scorer(request, ops, oracle, golden, syn):
    """Runs every scorer test under all three matrix-core modes."""
    ops.set_mma_mode({"bf16x6": ops.MMA_BF16X6, "f16x3": ops.MMA_F16X3, "f16x3l32": ops.MMA_F16X3_L32, "f32": ops.MMA_F32}[request.param])
    request.addfinalizer(lambda: ops.set_mma_mode(ops.MMA_DEFAULT))
    g = golden("g5_scorer")
    sd = syn.make_scorer_state_dict(0)
    rays = syn.make_rays(4096, 0)
    w = ops.PackedWeights({k: torch.from_numpy(v) for k, v in sd.items()}, "cuda")
    f16 = request.param in ("f16x3", "f16x3l32")
    feat, key, *rest = ops.ray_keys(G(rays["ori"]), G(rays["dir"]), G(rays["rgb"]), w, want_feat=True, want_planes=f16)
    ofeat, okey = oracle.ray_features(rays["ori"], rays["dir"], rays["rgb"], sd)
    # the f16x3 modes score through the DMA-fed kernels on pre-split planes; f32 and bf16x6 on the fp32 keys (key planes exist in ONE format since round 6)
    planes = kscale = None
    if not f16:
        with pytest.raises(RuntimeError, match="fp16 x 3 modes only"):
            ops.ray_keys(G(rays["ori"]), G(rays["dir"]), G(rays["rgb"]), w, want_planes=True)
    if f16:
        planes, kscale = rest[0]
        # the planes written chunk by chunk behind k_proj are those of a single split pass over the finished keys
        p2, s2 = ops.split_planes_f16(key)
        assert torch.equal(p2, planes) and torch.equal(s2, kscale)
        # without a feature output the f16x3 modes run the plane-to-plane chain (dense.hip): chunking must be invisible there too
        _, kfast, (p4, s4) = ops.ray_keys(G(rays["ori"]), G(rays["dir"]), G(rays["rgb"]), w, want_planes=True)
        _, _, (p3, s3) = ops.ray_keys(G(rays["ori"]), G(rays["dir"]), G(rays["rgb"]), w, want_key=False, want_planes=True, max_chunk=1024)
        assert torch.equal(p3, p4) and torch.equal(s3, s4)
        p5, s5 = ops.split_planes_f16(kfast)
        assert torch.equal(p5, p4) and torch.equal(s5, s4)
        assert rel_err(N(kfast), okey) < 5e-6                 # the chain's keys against the oracle ...
        assert rel_err(N(kfast), N(key)) < 2e-6               # ... and against the fp32-operand kernels' keys
    return dict(g=g, sd=sd, rays=rays, w=w, feat=feat, key=key, ofeat=ofeat, okey=okey, mode=request.param,
                planes=planes, kscale=kscale)


def test_a12_ray_encode(ops, oracle, scorer):
    r = scorer["rays"]
    x = N(ops.ray_encode(G(r["ori"]), G(r["dir"]), G(r["rgb"])))
    ox = oracle.ray_input(r["ori"], r["dir"], r["rgb"])
    assert x.shape == (4096, 144)
    assert np.abs(x[:, :141] - ox).max() < 5e-7          # |arg| up to ~500: sin/cos within an ulp or two
    assert (x[:, 141:] == 0).all()


@pytest.mark.parametrize("mode", ["bf16x6", "f32"])
def test_linear_mfma_vs_fp64(ops, mode):
    """Both tile kernels (fp32 MFMA chain; 3-plane bf16 split with 6 cross terms) against an fp64 product,
    with an asymmetric weight matrix (catches transposed fragments) and ragged M / K tails."""
    mm = ops.MMA_BF16X6 if mode == "bf16x6" else ops.MMA_F32
    rng = np.random.default_rng(0)
    for m, k, n in ((300, 144, 512), (128, 656, 512), (1000, 384, 384), (77, 20, 128), (129, 36, 256)):
        x = rng.standard_normal((m, k)).astype(np.float32)
        w = (rng.standard_normal((n, k)) * np.linspace(0.5, 2.0, n)[:, None]).astype(np.float32)
        b = rng.standard_normal(n).astype(np.float32)
        y = N(ops.linear(G(x), G(w), G(b), relu=False, mma_mode=mm))
        ref = x.astype(np.float64) @ w.astype(np.float64).T + b
        assert rel_err(y, ref) < 2e-6, (m, k, n)
        # error relative to sum |x||w| (the quantity fp32 rounding scales with): fp32-class for both modes
        den = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T + np.abs(b)
        assert (np.abs(y - ref) / den).max() < 4e-7, (mode, m, k, n)
        yr = N(ops.linear(G(x), G(w), G(b), relu=True, mma_mode=mm))
        assert rel_err(yr, np.maximum(ref, 0)) < 2e-6
    # wide dynamic range and tiny magnitudes survive the 3-way split (bf16 keeps the fp32 exponent range)
    x = (rng.standard_normal((256, 384)) * np.logspace(-6, 6, 256)[:, None]).astype(np.float32)
    w = (rng.standard_normal((128, 384)) * 1e-12).astype(np.float32)
    y = N(ops.linear(G(x), G(w), None, mma_mode=mm))
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    den = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T
    assert (np.abs(y - ref) / den).max() < 4e-7


def test_a13_ray_features_and_keys(scorer):
    g = scorer["g"]
    assert rel_err(N(scorer["feat"]), scorer["ofeat"]) < 5e-6
    assert rel_err(N(scorer["key"]), scorer["okey"]) < 5e-6
    assert rel_err(N(scorer["feat"])[:128], g["feat_head"]) < 5e-6
    assert rel_err(N(scorer["key"])[:128], g["key_head"]) < 5e-6


def test_ray_keys_chunked_equals_unchunked(ops, scorer):
    r = scorer["rays"]
    ws = torch.empty(1000 * 1580 * 4, dtype=torch.uint8, device="cuda")   # forces ~900-ray chunks (896: whole 128-ray scale tiles)
    _, key2 = ops.ray_keys(G(r["ori"]), G(r["dir"]), G(r["rgb"]), scorer["w"], workspace=ws, max_chunk=1000)
    _, key1 = ops.ray_keys(G(r["ori"]), G(r["dir"]), G(r["rgb"]), scorer["w"])
    assert torch.equal(key2, key1)
    assert rel_err(N(key2), scorer["okey"]) < 5e-6
    fb, kb = ops.ray_keys(G(r["ori"]), G(r["dir"]), G(r["rgb"]), scorer["w"], workspace=ws, max_chunk=1000, want_feat=True)
    assert torch.equal(kb, scorer["key"]) and torch.equal(fb, scorer["feat"])


@pytest.mark.parametrize("tag,T,scale", [("flat256", 256, 1.0), ("peaky256", 256, 40.0), ("peaky137", 137, 40.0),
                                         ("mid1", 1, 10.0)])
def test_a14_a15_score_topk(ops, oracle, scorer, syn, tag, T, scale):
    g = scorer["g"]
    tok = syn.make_tokens(T, 1, scale)
    tokens, n_tok = ops.pad_tokens([G(tok)], "cuda")
    q = ops.q_proj(tokens, n_tok, scorer["w"])
    oq = oracle.q_proj(tok, scorer["sd"])
    assert rel_err(N(q)[0, :T], oq) < 5e-6
    assert (N(q)[0, T:] == 0).all()
    idx, val, scores, stats = ops.score_topk(q, n_tok, scorer["key"], 100, want_stats=True, key_planes=scorer["planes"],
                                             key_scale=scorer["kscale"])
    s = N(scores)[0]
    # values: 1e-5 relative against the reference's fp32 result and against the oracle
    assert rel_err(s, g[f"{tag}_scores"]) < 1e-5
    os_, omx, osm = oracle.attention_scores(oq, scorer["okey"], return_stats=True)
    assert rel_err(s, os_) < 1e-5
    assert np.abs(N(stats)[0, :T, 0] - omx).max() < 1e-4
    assert rel_err(N(stats)[0, :T, 1], osm) < 1e-5
    assert abs(float(s.astype(np.float64).sum()) - T) < 1e-3 * T
    # top-k: exactly the kernel's own scores sorted (value desc, index asc) ...
    order = np.lexsort((np.arange(s.size), -s))[:100]
    assert (N(idx)[0] == order).all()
    assert (N(val)[0] == s[order]).all()
    # ... and the reference's index set wherever the fp64 gap exceeds 4x the fp32 error bound
    s64 = g[f"{tag}_scores64"]
    o64 = np.argsort(-s64)
    must = set(o64[:100][s64[o64[:100]] - s64[o64[100]] > 4e-6 * s64[o64[0]]].tolist())
    assert must <= set(N(idx)[0].tolist())
    assert set(N(idx)[0].tolist()) == set(g[f"{tag}_idx"].tolist())
    if tag != "flat256":
        assert (N(idx)[0] == g[f"{tag}_idx"]).all()      # order too when the gaps are real


def test_score_topk_batched_and_grouped(ops, scorer, syn):
    """A batch with ragged token counts, processed (a) all at once and (b) one image at a time through a
    workspace that only fits a single image -- identical results."""
    toks = [syn.make_tokens(t, 10 + i, 40.0) for i, t in enumerate((256, 137, 1, 200, 0))]
    tokens, n_tok = ops.pad_tokens([G(t) for t in toks], "cuda")
    q = ops.q_proj(tokens, n_tok, scorer["w"])
    kp, ks = scorer["planes"], scorer["kscale"]
    idx, val, sc, _ = ops.score_topk(q, n_tok, scorer["key"], 100, key_planes=kp, key_scale=ks)
    idx1, val1, sc1, _ = ops.score_topk(q, n_tok, scorer["key"], 100, images_in_flight=1, key_planes=kp, key_scale=ks)
    assert torch.equal(idx, idx1) and torch.equal(val, val1) and torch.equal(sc, sc1)
    for i in range(4):
        qi = q[i:i + 1].contiguous()
        ii, vi, si, _ = ops.score_topk(qi, n_tok[i:i + 1].contiguous(), scorer["key"], 100, key_planes=kp, key_scale=ks)
        assert torch.equal(ii[0], idx[i]) and torch.equal(si[0], sc[i])
    assert (N(sc)[4] == 0).all()                          # no tokens -> all-zero scores (empty sum)
    assert (N(idx)[4] == np.arange(100)).all()            # all ties -> lowest indices


def test_scorer_kernels_agree_and_fp16_planes_are_exact(ops, scorer, syn):
    """The logits kernels (fp32 MFMA chain and bf16x6 with the split on the fly, both on fp32 keys; fp16 x 3 DMA-fed on scaled planes, with 24-bit and
    fp32 logits) agree to fp32 rounding and in their top-100; the scaled fp16 planes reproduce the keys to 2^-22 of each tile's maximum; key planes
    handed to a mode that scores on fp32 keys are ignored when `key` is there and refused when it is not."""
    key = scorer["key"]
    tok = syn.make_tokens(256, 5, 40.0)
    tokens, n_tok = ops.pad_tokens([G(tok)], "cuda")
    q = ops.q_proj(tokens, n_tok, scorer["w"])
    res = {}
    p16, s16 = ops.split_planes_f16(key)
    for name, mode, kp, ks in (("f32", ops.MMA_F32, None, None), ("b6", ops.MMA_BF16X6, None, None),
                               ("b6planes", ops.MMA_BF16X6, p16, s16), ("f16x3", ops.MMA_F16X3, p16, s16),
                               ("f16x3l32", ops.MMA_F16X3_L32, p16, s16)):
        ops.set_mma_mode(mode)
        res[name] = ops.score_topk(q, n_tok, key, 100, key_planes=kp, key_scale=ks)
    for a, b in zip(res["b6"], res["b6planes"]):           # planes next to fp32 keys in a mode that scores on the keys: ignored
        assert a is None or torch.equal(a, b)
    ops.set_mma_mode(ops.MMA_BF16X6)
    with pytest.raises(RuntimeError, match="fp32 keys"):
        ops.score_topk(q, n_tok, None, 100, key_planes=p16, key_scale=s16)
    ops.set_mma_mode({"bf16x6": ops.MMA_BF16X6, "f16x3": ops.MMA_F16X3, "f16x3l32": ops.MMA_F16X3_L32, "f32": ops.MMA_F32}[scorer["mode"]])
    # the scaled fp16 planes reproduce the keys to 2^-22 relative to the largest key of each 128-ray tile
    kn = N(key).astype(np.float64)
    inv = N(s16).astype(np.float64)
    assert inv.shape[0] == (kn.shape[0] + 127) // 128 and np.all(np.log2(inv) == np.round(np.log2(inv)))
    h = p16.cpu().numpy().reshape(-1, 12, 2, 32, 2)
    rec = (h[..., 0].astype(np.uint16) | (h[..., 1].astype(np.uint16) << 8)).view(np.float16).astype(np.float64).sum(axis=2).reshape(-1, 384)
    for t in range(inv.shape[0]):
        blk = kn[t * 128:(t + 1) * 128]
        assert np.abs(rec[t * 128:(t + 1) * 128] * inv[t] - blk).max() <= 2.0 ** -22 * np.abs(blk).max()
        assert 2.0 ** 13 <= np.abs(blk).max() / inv[t] < 2.0 ** 14
    # the 24-bit logits change no bit of the row statistics and move the scores by < 1e-6
    assert float((res["f16x3"][2] - res["f16x3l32"][2]).abs().max() / res["f16x3l32"][2].abs().max()) < 1e-6
    for name in ("b6", "f16x3", "f16x3l32"):
        assert rel_err(N(res[name][2]), N(res["f32"][2])) < 5e-6, name
        assert (N(res[name][0]) == N(res["f32"][0])).all(), name


@pytest.mark.parametrize("r", [1, 37, 129, 300])
def test_scorer_on_tiny_ray_sets(ops, oracle, r):
    """Fewer rays than a tile / than k: scores against the oracle, top-k padded with (-1, NaN) beyond the r-th entry."""
    rng = np.random.default_rng(r)
    key = rng.standard_normal((r, 384)).astype(np.float32)
    q = np.zeros((2, 256, 384), np.float32)
    q[0, :200] = rng.standard_normal((200, 384)).astype(np.float32) * 0.3
    q[1, :3] = rng.standard_normal((3, 384)).astype(np.float32) * 0.3
    n_tok = torch.tensor([200, 3], dtype=torch.int32, device="cuda")
    for mode in (ops.MMA_F16X3, ops.MMA_BF16X6, ops.MMA_F32):
        ops.set_mma_mode(mode)
        try:
            kp, ks = ops.split_planes_f16(G(key)) if mode == ops.MMA_F16X3 else (None, None)
            idx, val, sc, _ = ops.score_topk(G(q), n_tok, G(key), 100, key_planes=kp, key_scale=ks)
        finally:
            ops.set_mma_mode(ops.MMA_DEFAULT)
        for b, t in enumerate((200, 3)):
            ref = oracle.attention_scores(q[b, :t], key)
            assert np.abs(N(sc)[b] - ref).max() <= 1e-5 * np.abs(ref).max(), (mode, b)
            assert abs(float(N(sc)[b].astype(np.float64).sum()) - t) < 1e-4 * t
            n = min(r, 100)
            order = np.lexsort((np.arange(r), -N(sc)[b]))[:n]
            assert (N(idx)[b, :n] == order).all() and (N(idx)[b, n:] == -1).all() and np.isnan(N(val)[b, n:]).all()


@pytest.mark.parametrize("m,k,n", [(576, 9600, 384), (4, 6144, 384), (64, 9600, 384), (130, 2052, 37)])
def test_linear_split_k_matches_single_pass(ops, m, k, n):
    """Split-K (few output tiles, long K: the camera-up CNN as im2col GEMMs) adds the K slices in a fixed order: deterministic,
    and equal to the single-pass product up to the re-association of the fp32 sum."""
    rng = np.random.default_rng(m + n)
    x = G(rng.standard_normal((m, k)).astype(np.float32))
    w = G((rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32))
    b = G(rng.standard_normal(n).astype(np.float32))
    ref = (N(x).astype(np.float64) @ N(w).astype(np.float64).T + N(b)).clip(min=0)
    y1 = ops.linear(x, w, b, relu=True, split_k=1)
    for s in (None, 7, 32):
        ys = ops.linear(x, w, b, relu=True, split_k=s)
        assert torch.equal(ys, ops.linear(x, w, b, relu=True, split_k=s))            # deterministic
        assert np.abs(N(ys) - ref).max() < 3e-6 * max(1.0, np.abs(ref).max())
    assert np.abs(N(y1) - ref).max() < 3e-6 * max(1.0, np.abs(ref).max())


def test_f16x3_tile_scaling_edge_cases(ops):
    """fp16x3 scorer on operands whose 128-row tiles differ by many orders of magnitude, with an all-zero tile, a ragged
    last tile and values far outside the fp16 range: the per-token max logit stays within the fp32 error bound of the
    float64 result, as tight as the fp32 MFMA chain."""
    rng = np.random.default_rng(7)
    R, T = 128 * 9 + 37, 200
    key = rng.standard_normal((R, 384)).astype(np.float32)
    tile_scale = np.array([1e-20, 1e-6, 1.0, 0.0, 3e4, 1e9, 2.0 ** -30, 7.0, 1e-3, 1e12], dtype=np.float32)
    key *= np.repeat(tile_scale, 128)[:R, None]
    q = np.zeros((1, 256, 384), dtype=np.float32)
    q[0, :T] = rng.standard_normal((T, 384)).astype(np.float32)
    q[0, :128] *= 1e-10                                     # the two token halves get their own scale
    q[0, 128:T] *= 3e-12
    ref = (q[0, :T].astype(np.float64) @ key.astype(np.float64).T) / np.sqrt(384.0)           # [T,R]
    bound = (np.abs(q[0, :T]).astype(np.float64) @ np.abs(key).astype(np.float64).T) / np.sqrt(384.0)
    n_tok = torch.tensor([T], dtype=torch.int32, device="cuda")
    planes, inv = ops.split_planes_f16(G(key))
    assert float(inv[3]) == 1.0                             # all-zero tile: scale 1
    err = {}
    for name, mode, kp, ks in (("f32", ops.MMA_F32, None, None), ("f16x3", ops.MMA_F16X3, planes, inv)):
        ops.set_mma_mode(mode)
        try:
            _, _, _, stats = ops.score_topk(G(q), n_tok, G(key), 100, want_stats=True, key_planes=kp, key_scale=ks)
        finally:
            ops.set_mma_mode(ops.MMA_DEFAULT)
        mx = N(stats)[0, :T, 0].astype(np.float64)
        j = ref.argmax(axis=1)
        err[name] = np.abs(mx - ref[np.arange(T), j]) / bound[np.arange(T), j]
    assert err["f32"].max() < 4e-7
    assert err["f16x3"].max() < 4e-7


def test_topk_ties_short_and_large(ops, oracle):
    rng = np.random.default_rng(1)
    # heavy ties
    s = rng.integers(0, 7, size=(3, 5000)).astype(np.float32)
    idx, val = ops.topk(G(s), 100)
    for b in range(3):
        oi, ov = oracle.topk(s[b], 100)
        assert (N(idx)[b] == oi).all() and (N(val)[b] == ov).all()
    # fewer elements than k: padded with (-1, NaN)
    s = rng.standard_normal((2, 37)).astype(np.float32)
    idx, val = ops.topk(G(s), 100)
    for b in range(2):
        oi, ov = oracle.topk(s[b], 100)
        assert (N(idx)[b, :37] == oi).all() and (N(idx)[b, 37:] == -1).all() and np.isnan(N(val)[b, 37:]).all()
    # negative, zero, large; multi-block
    s = (rng.standard_normal((2, 300_000)) * 5).astype(np.float32)
    s[0, 123456] = np.inf
    idx, val = ops.topk(G(s), 100)
    for b in range(2):
        oi, ov = oracle.topk(s[b], 100)
        assert (N(idx)[b] == oi).all() and (N(val)[b] == ov).all()


# ------------------------------------------------------------------------------------------------
# pose tail
# ------------------------------------------------------------------------------------------------
def test_a17_to_a21_solve_pose(ops, oracle, golden):
    g = golden("g6_pose")
    cases = [str(c) for c in g["cases"]]
    for name in cases:
        k = g[f"{name}_idx"].shape[0]
        out = ops.solve_pose(G(g[f"{name}_ori"]), G(g[f"{name}_dir"]), G(g[f"{name}_idx"])[None], G(g[f"{name}_w"])[None],
                             G(g[f"{name}_up"])[None], G(g[f"{name}_gt"])[None])
        keep = N(out["w_final"])[0] != 0
        o = oracle.pose_from_topk(g[f"{name}_ori"], g[f"{name}_dir"], g[f"{name}_idx"], g[f"{name}_w"], g[f"{name}_up"])
        assert int(out["n_kept"][0]) == int(g[f"{name}_keep_mask"].sum()), name
        st = int(out["status"][0])
        assert bool(st & 1) == bool(g[f"{name}_flags"][0]) and bool(st & 2) == bool(g[f"{name}_flags"][1]), name
        assert np.abs(N(out["c2w"])[0] - g[f"{name}_c2w"]).max() < 1e-5, name
        assert np.abs(N(out["c2w"])[0] - o["c2w"]).max() < 1e-5, name
        if np.isnan(g[f"{name}_centre"]).any():
            assert st & 4 and np.isnan(N(out["centre"])[0]).all()
        else:
            assert np.abs(N(out["centre"])[0] - g[f"{name}_centre"]).max() < 1e-5, name
            # dropped rays have weight 0; kept-but-excluded ones too, so compare the dense vector
            assert np.abs(N(out["w_final"])[0] - o["w_final"]).max() < 1e-7, name
        assert abs(float(out["errors"][0, 0]) - float(g[f"{name}_terr"])) < 1e-5, name
        assert abs(float(out["errors"][0, 1]) - float(g[f"{name}_aerr"])) < 1e-3, name
        assert k <= 256 and keep.sum() <= k


def test_solve_pose_batched_equals_single(ops, golden):
    g = golden("g6_pose")
    names = ["plain", "dups", "behind", "mixed"]
    R = g["plain_ori"].shape[0]
    # stack the four ray sets into one array and offset the indices
    ori = np.concatenate([g[f"{n}_ori"] for n in names])
    dr = np.concatenate([g[f"{n}_dir"] for n in names])
    idx = np.stack([g[f"{n}_idx"] + i * R for i, n in enumerate(names)])
    w = np.stack([g[f"{n}_w"] for n in names])
    up = np.stack([g[f"{n}_up"] for n in names])
    out = ops.solve_pose(G(ori), G(dr), G(idx), G(w), G(up))
    for i, n in enumerate(names):
        assert np.abs(N(out["c2w"])[i] - g[f"{n}_c2w"]).max() < 1e-5
    assert np.isnan(N(out["errors"])).all()               # no ground truth given


def test_topk_randomised_against_the_oracle(ops, oracle):
    """Radix select + ordered gather + bitonic sort against the oracle's (value desc, index asc) rule on 60 random shapes:
    heavy ties, all-equal rows, +-inf, signed zeros, sizes around the block / pass boundaries, k from 1 to 1024."""
    rng = np.random.default_rng(2024)
    for trial in range(60):
        r = int(rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 1000, 1023, 1024, 1025, 4095, 4096, 4097, 65537, 131072, 131073, 300001]))      # <= 2^17: the one-workgroup kernel (k_topk_small); beyond: the multi-kernel path
        k = int(rng.choice([1, 7, 100, 256, 1024]))
        kind = trial % 5
        if kind == 0:
            s = rng.standard_normal((2, r))
        elif kind == 1:
            s = rng.integers(-3, 4, size=(2, r)).astype(np.float64)             # heavy ties, both zeros' signs below
        elif kind == 2:
            s = np.full((2, r), 0.25)
        elif kind == 3:
            s = rng.standard_normal((2, r)) * 1e-30                              # denormal-ish magnitudes
        else:
            s = rng.standard_normal((2, r))
            s[0, rng.integers(0, r, size=min(r, 5))] = np.inf
            s[1, rng.integers(0, r, size=min(r, 5))] = -np.inf
        s = s.astype(np.float32)
        if kind == 1:
            s[0, ::3] *= np.float32(-1.0)                                         # -0.0 among the zeros: equal to +0.0
        idx, val = ops.topk(G(s), k)
        for b in range(2):
            oi, ov = oracle.topk(s[b], k)
            n = min(k, r)
            assert (N(idx)[b, :n] == oi[:n]).all(), (trial, r, k, kind)
            assert (N(val)[b, :n] == ov[:n]).all() and (N(idx)[b, n:] == -1).all()


def test_pose_tail_randomised_against_the_oracle(ops, oracle):
    """a17-a21 on 60 random top-k sets (duplicate origins in varying multiplicity, rays behind the centre, 3 <= k <= 100):
    kept-ray count, status flags, centre, final weights and c2w against the oracle (itself pinned by the g6 goldens)."""
    rng = np.random.default_rng(77)
    R = 500
    for trial in range(60):
        k = int(rng.choice([3, 5, 8, 17, 37, 100]))
        c = rng.standard_normal(3).astype(np.float32) * 2
        ori = rng.standard_normal((R, 3)).astype(np.float32)
        dr = c[None] - ori + rng.standard_normal((R, 3)).astype(np.float32) * 0.05
        dr /= np.linalg.norm(dr, axis=1, keepdims=True)
        flip = rng.random(R) < 0.15
        dr[flip] *= -1                                                  # camera behind those rays
        n_dup = int(rng.integers(0, 4))
        for _ in range(n_dup):                                          # duplicated origins (several rays of one ellipsoid point)
            a, b = rng.integers(0, R, size=2)
            ori[b] = ori[a]
        idx = rng.choice(R, size=k, replace=False).astype(np.int64)
        w = np.sort(rng.random(k).astype(np.float32))[::-1].copy()
        up = rng.standard_normal(3).astype(np.float32)
        up /= np.linalg.norm(up)
        out = ops.solve_pose(G(ori), G(dr), G(idx)[None], G(w)[None], G(up)[None])
        o = oracle.pose_from_topk(ori, dr, idx, w, up)
        assert int(out["n_kept"][0]) == o["n_kept"], trial
        st = int(out["status"][0])
        assert bool(st & 1) == bool(o["flags"][0]) and bool(st & 2) == bool(o["flags"][1]), trial
        if np.isnan(o["centre"]).any():
            assert st & 4
            continue
        scale = max(1.0, float(np.abs(o["c2w"]).max()))
        assert np.abs(N(out["centre"])[0] - o["centre"]).max() <= 2e-4 * scale, trial     # 3x3 solves of near-parallel ray bundles
        assert np.abs(N(out["w_final"])[0] - o["w_final"]).max() < 1e-6, trial
        assert np.abs(N(out["c2w"])[0] - o["c2w"]).max() <= 2e-4 * scale, trial


def test_geometry_randomised_against_the_oracle(ops, oracle):
    """Wide-range random inputs for the integer-valued geometry rows: validity mask and ring / cell counts on ellipsoids whose
    axes span 1e-4 .. 1e2 with strong anisotropy (a2, a6), closed-form eigen-decomposition of random, nearly degenerate and
    exactly degenerate symmetric matrices (a5), iso-cell distributions for 20 targets (a8)."""
    rng = np.random.default_rng(99)
    # a2 / a6: scales log-uniform over six decades
    scale = np.exp(rng.uniform(np.log(1e-4), np.log(1e2), size=(600, 3))).astype(np.float32)
    for P in (17, 50, 64):
        m = N(ops.mask_degraded(G(np.log(scale)), P))
        act = np.exp(np.log(scale).astype(np.float32)).astype(np.float32)
        ref = oracle.mask_degraded(act, P)
        assert (m != ref).sum() <= 2                                   # exp(log s) on the GPU may differ from the host's by an ulp
        ok = ref & m
        sub = act[ok][:200]
        pts, eid = ops.quadricell_centers(G(sub), P)
        opts, oeid = oracle.quadricell_centers(sub, P)
        assert pts.shape[0] == opts.shape[0] and (N(eid) == oeid).all()   # identical ring and cell counts for every ellipsoid
        d = np.abs(N(pts) - opts).max(1) / sub[oeid].max(1)
        assert np.quantile(d, 0.98) < 1e-5                              # the rest: table ties (conftest.quadricell_tie_cells)
    # a5: random symmetric, near-degenerate (two close eigenvalues), exactly diagonal, rank one
    a = rng.standard_normal((300, 3, 3))
    mats = [a @ a.transpose(0, 2, 1)]
    q, _ = np.linalg.qr(rng.standard_normal((100, 3, 3)))
    lam = np.stack([np.ones(100), 1 + 1e-6 * rng.random(100), 3 * np.ones(100)], 1)
    mats.append(q @ (lam[:, :, None] * q.transpose(0, 2, 1)))
    mats.append(np.stack([np.diag(v) for v in rng.random((50, 3))]))
    v = rng.standard_normal((50, 3, 1))
    mats.append(v @ v.transpose(0, 2, 1))
    mats = np.concatenate(mats).astype(np.float32)
    vals, vecs = ops.sym_eig_3x3(G(mats))
    ov, oe = oracle.sym_eig_3x3(mats)
    sc = np.abs(ov).max(1, keepdims=True) + 1e-30
    assert (np.abs(N(vals) - ov) / sc).max() < 5e-5
    w = np.linalg.eigvalsh(mats.astype(np.float64))
    assert (np.abs(np.sort(N(vals), axis=1) - w) / sc).max() < 2e-4     # and they ARE the eigenvalues
    gap = np.minimum(ov[:, 1] - ov[:, 0], ov[:, 2] - ov[:, 1]) / sc[:, 0]
    ok = gap > 1e-2
    assert np.abs(N(vecs)[ok] - oe[ok]).max() < 5e-3
    # a8: iso-cell directions
    for tgt in (1, 2, 3, 7, 10, 16, 35, 50, 64, 100, 128, 200, 256, 300, 500, 777, 1000, 1024, 2048, 4096):
        for n0 in (1, 3):
            d = N(ops.isocell_distribution(tgt, n0))
            od = oracle.isocell_distribution(tgt, n0)
            assert d.shape == od.shape and np.abs(d - od).max() < 2e-6, (tgt, n0)


def test_plane_chain_keys_on_wide_dynamic_range_inputs(ops, oracle, syn):
    """The plane-to-plane ray MLP chain (scaled fp16 x 3, one scale per ray and block of 128 features) on rays whose coordinates span
    six orders of magnitude and on a ragged ray count: keys against the oracle's fp32 chain."""
    sd = syn.make_scorer_state_dict(3)
    w = ops.PackedWeights({k: torch.from_numpy(v) for k, v in sd.items()}, "cuda")
    rays = syn.make_rays(5003, 9)
    rays["ori"] = (rays["ori"] * np.logspace(-3, 3, 5003)[:, None]).astype(np.float32)
    rays["rgb"][::7] = 0.0
    _, key = ops.ray_keys(G(rays["ori"]), G(rays["dir"]), G(rays["rgb"]), w)
    _, okey = oracle.ray_features(rays["ori"], rays["dir"], rays["rgb"], sd)
    assert bool(torch.isfinite(key).all())
    # row-wise: every key row within 5e-6 of its own largest element
    err = np.abs(N(key) - okey).max(axis=1) / np.abs(okey).max(axis=1)
    assert err.max() < 5e-6, err.max()


def test_plane_chain_persistent_workgroups_over_many_tiles(ops, oracle, syn):
    """More ray tiles than compute units: every workgroup of k_dense_planes walks over several tiles (tile switch, the next tile's input
    shifts, the load cursor running ahead into the next tile) and the last tile is ragged.  Keys of the first, a middle and the last
    stretch of rays against the oracle's fp32 chain, and the whole set against a second run in small chunks (one tile per workgroup)."""
    sd = syn.make_scorer_state_dict(5)
    w = ops.PackedWeights({k: torch.from_numpy(v) for k, v in sd.items()}, "cuda")
    R = 256 * 256 * 3 + 4321                      # > 3 tiles of 256 rays (and > 6 of 128) per workgroup on 256 CUs
    rays = syn.make_rays(R, 21)
    rays["ori"] = (rays["ori"] * np.logspace(-2, 2, R)[:, None]).astype(np.float32)
    o, d, c = G(rays["ori"]), G(rays["dir"]), G(rays["rgb"])
    _, key = ops.ray_keys(o, d, c, w)
    assert key.shape == (R, 384) and bool(torch.isfinite(key).all())
    for lo, hi in ((0, 700), (R // 2 - 300, R // 2 + 300), (R - 4321 - 200, R)):
        _, okey = oracle.ray_features(rays["ori"][lo:hi], rays["dir"][lo:hi], rays["rgb"][lo:hi], sd)
        err = np.abs(N(key[lo:hi]) - okey).max(axis=1) / np.abs(okey).max(axis=1)
        assert err.max() < 5e-6, (lo, err.max())
    ws = torch.empty(20000 * 1580 * 4, dtype=torch.uint8, device="cuda")
    _, key2 = ops.ray_keys(o, d, c, w, workspace=ws, max_chunk=20000)
    assert torch.equal(key2, key)


def test_k_proj_writes_the_key_planes_itself(ops, syn):
    """want_key=False, want_planes=True (the key cache): the last layer of the plane chain emits the scorer's tile-scaled fp16 key planes and
    the reciprocal tile scales directly (one workgroup tile of 128 rays x 384 features is one key tile).  Bit-identical to splitting the fp32
    keys with sixdgs_split_planes_f16, including a ragged last tile, chunked runs and a tile of zeros."""
    sd = syn.make_scorer_state_dict(11)
    w = ops.PackedWeights({k: torch.from_numpy(v) for k, v in sd.items()}, "cuda")
    prev = ops.get_mma_mode()
    ops.set_mma_mode(ops.MMA_F16X3)
    try:
        _k_proj_planes_cases(ops, syn, w)
    finally:
        ops.set_mma_mode(prev)


def _k_proj_planes_cases(ops, syn, w):
    for R, chunk in ((128 * 301 + 77, 262144), (128 * 301 + 77, 128 * 90), (5, 262144), (128, 262144)):
        rays = syn.make_rays(R, 31)
        rays["ori"] = (rays["ori"] * np.logspace(-2, 3, R)[:, None]).astype(np.float32)
        o, d, c = G(rays["ori"]), G(rays["dir"]), G(rays["rgb"])
        _, key, (planes_a, inv_a) = ops.ray_keys(o, d, c, w, want_key=True, want_planes=True, max_chunk=chunk)       # fp32 keys + split kernel
        _, none, (planes_b, inv_b) = ops.ray_keys(o, d, c, w, want_key=False, want_planes=True, max_chunk=chunk)    # fused
        assert none is None
        assert torch.equal(inv_a, inv_b), (R, chunk)
        assert torch.equal(planes_a, planes_b), (R, chunk)
        # the planes decode to the keys: (h + l) * inv_scale within 2^-21 of the tile's largest magnitude
        pl = planes_b.view(torch.float16).view(R, 12, 2, 32).float()
        dec = (pl[:, :, 0] + pl[:, :, 1]).reshape(R, 384) * inv_b.repeat_interleave(128)[:R, None]
        tile_max = torch.zeros((R + 127) // 128, device="cuda").scatter_reduce(0, torch.arange(R, device="cuda") // 128, key.abs().amax(dim=1), "amax")
        assert float(((dec - key).abs().amax(dim=1) / tile_max.repeat_interleave(128)[:R]).max()) < 2.0 ** -20


def test_plane_chain_randomised_sizes_and_chunks(ops, syn):
    """Ray counts around the tile sizes of both dense kernels (128 / 256 rays), the persistent grid (256 workgroups) and random chunkings:
    the fused key planes equal the split of the fp32 keys, and every chunking gives the same keys bit for bit."""
    sd = syn.make_scorer_state_dict(17)
    w = ops.PackedWeights({k: torch.from_numpy(v) for k, v in sd.items()}, "cuda")
    rng = np.random.default_rng(123)
    sizes = [1, 127, 128, 129, 255, 256, 257, 383, 32768 + 5, 65536 + 1, 256 * 257 - 3] + [int(x) for x in rng.integers(2, 90000, 6)]
    prev = ops.get_mma_mode()
    ops.set_mma_mode(ops.MMA_F16X3)
    try:
        for R in sizes:
            rays = syn.make_rays(R, 40 + R % 7)
            o, d, c = G(rays["ori"]), G(rays["dir"]), G(rays["rgb"])
            _, key, (pa, ia) = ops.ray_keys(o, d, c, w, want_key=True, want_planes=True)
            chunk = int(rng.integers(1, 40)) * 128
            _, key2, (pb, ib) = ops.ray_keys(o, d, c, w, want_key=True, want_planes=True, max_chunk=chunk)
            _, _, (pc, ic) = ops.ray_keys(o, d, c, w, want_key=False, want_planes=True, max_chunk=chunk)
            assert torch.equal(key, key2), (R, chunk)
            assert torch.equal(pa, pb) and torch.equal(ia, ib), (R, chunk)
            assert torch.equal(pa, pc) and torch.equal(ia, ic), (R, chunk)
            assert bool(torch.isfinite(key).all())
    finally:
        ops.set_mma_mode(prev)


def test_chain_layouts_give_identical_keys(tmp_path):
    """The plane-to-plane chain keeps its activations chunk-major (default) or ray-major (SIXDGS_DENSE_CM=0; one layout per process, the weight
    planes are packed for it).  Same MFMA order per output and same scales: the key planes, tile scales, fp32 keys and the key-norm maximum of
    the two are bit-identical, ragged tiles and chunked runs included (tools/cm_check.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    f = str(tmp_path / "ray_major.pt")
    for mode, args in (("0", ["save", f]), ("1", ["compare", f])):
        p = subprocess.run([sys.executable, "-W", "ignore", os.path.join(root, "tools", "cm_check.py"), *args], env=dict(os.environ, SIXDGS_DENSE_CM=mode),
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-2000:])
    assert "CM_CHECK PASS" in p.stdout


def test_topk_small_lists_equal_the_multi_kernel_path(tmp_path):
    """Round 4: lists of <= 2^17 values are selected by ONE workgroup per image (k_topk_small) instead of nine launches.  Same definition, same
    bits: the tool runs both paths (SIXDGS_TOPK_SMALL=0 forces the multi-kernel one; read once per process) on lists with heavy ties, all-equal rows,
    infinities and sizes around 1024 / 2^17, and compares idx and val exactly."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = (
        "import importlib, sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "ops = importlib.import_module('6dgs_amd.ops')\n"
        "rng = np.random.default_rng(77); out = {}\n"
        "for r in (1, 37, 100, 101, 1023, 1024, 1025, 4096, 75000, 131071, 131072):\n"
        "    for kind in range(4):\n"
        "        s = [rng.standard_normal((3, r)), rng.integers(-2, 3, size=(3, r)).astype(np.float64), np.full((3, r), -1.5), rng.standard_normal((3, r)) * 1e-3][kind].astype(np.float32)\n"
        "        if kind == 0 and r > 40: s[0, 17] = np.inf; s[1, 33] = -np.inf\n"
        "        for k in (1, 100, 1024):\n"
        "            i, v = ops.topk(torch.from_numpy(s).cuda(), k)\n"
        "            out[f'{r}_{kind}_{k}'] = (i.cpu(), v.cpu())\n"
        "torch.save(out, sys.argv[1])\n" % root)
    files = []
    for mode in ("1", "0"):
        f = str(tmp_path / f"topk_{mode}.pt")
        p = subprocess.run([sys.executable, "-W", "ignore", "-c", prog, f], env=dict(os.environ, SIXDGS_TOPK_SMALL=mode), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        files.append(torch.load(f))
    a, b = files
    assert a.keys() == b.keys() and len(a) == 11 * 4 * 3
    for key in a:
        assert torch.equal(a[key][0], b[key][0]), key
        assert torch.equal(torch.nan_to_num(a[key][1], nan=-7.0), torch.nan_to_num(b[key][1], nan=-7.0)), key
