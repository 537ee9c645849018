"""Size-independent properties at the headline size of BASELINE.json (500 k Gaussians x 64 iso-cell rays = 32 M rays,
256 tokens): the oracle cannot run this size in seconds, so the checks are invariants of the path itself --
ray count / unit directions / hemisphere, softmax mass (sum of the scores of an image = its token count), top-k
consistency with the full score vector, batch invariance, and agreement of two independent arithmetic schemes
(fp16x3 and bf16x6) on the top-100 indices.  One scene build for the whole module (~5 s on MI355X)."""
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N_GAUSS, K_RAYS = 500_000, 64


@pytest.fixture(scope="module")
def full():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if torch.cuda.get_device_properties(0).total_memory < 200 * 2**30:
        pytest.skip("needs the 288 GB of an MI355X")
    pkg = importlib.import_module("6dgs_amd")
    syn = importlib.import_module("6dgs_amd.synthetic")
    ops = importlib.import_module("6dgs_amd.ops")
    ops.set_mma_mode(ops.MMA_DEFAULT)
    scene = pkg.GaussianScene.from_dict(syn.make_scene(N_GAUSS, 0), device="cuda")
    ori, dr, rgb, src = pkg.generate_all_possible_rays(scene, max_ellipsoids=-1, emitter="isocell", rays_per_ellipsoid=K_RAYS,
                                                       return_src=True)
    w = ops.PackedWeights({k: torch.from_numpy(v) for k, v in syn.make_scorer_state_dict(0).items()}, "cuda")
    toks = [syn.make_tokens(t, 20 + i, 40.0) for i, t in enumerate((256, 173))]
    tokens, n_tok = ops.pad_tokens([torch.from_numpy(t).cuda() for t in toks], "cuda")
    q = ops.q_proj(tokens, n_tok, w)
    d = dict(pkg=pkg, ops=ops, scene=scene, ori=ori, dr=dr, rgb=rgb, src=src, w=w, q=q, n_tok=n_tok)
    yield d
    d.clear()
    torch.cuda.empty_cache()


def test_emission_invariants_at_full_size(full):
    ori, dr, rgb, src, scene = full["ori"], full["dr"], full["rgb"], full["src"], full["scene"]
    n_valid = int(full["ops"].mask_degraded(scene._scaling, 50).sum())
    assert ori.shape == dr.shape == rgb.shape == (n_valid * K_RAYS, 3) and n_valid > 0.99 * N_GAUSS
    fin = torch.isfinite(dr).all(dim=1)
    assert float((~fin).float().mean()) < 1e-5                      # NaN only for normals exactly along +-z (isocell.py:208-212)
    nrm = dr[fin].norm(dim=1)
    assert float((nrm - 1).abs().max()) < 1e-5
    assert float(rgb.min()) >= 0.0                                   # clamp_min(rgb + 0.5, 0)
    # K rays per source ellipsoid, grouped, sources ascending
    assert torch.equal(src[::K_RAYS], src[K_RAYS - 1::K_RAYS]) and bool((src[K_RAYS::K_RAYS] > src[:-K_RAYS:K_RAYS]).all())
    # iso-cell rays leave the surface on the side of the normal: all K directions of an ellipsoid lie in one hemisphere
    d3 = torch.nan_to_num(dr.reshape(-1, K_RAYS, 3)[:4096])
    axis = torch.nn.functional.normalize(d3.mean(dim=1), dim=1)
    assert float((d3 * axis[:, None]).sum(-1).min()) > -1e-4


def test_scorer_invariants_at_full_size(full):
    ops, q, n_tok = full["ops"], full["q"], full["n_tok"]
    fin = torch.isfinite(full["dr"]).all(dim=1)
    ori, dr, rgb = (full[k][fin].contiguous() for k in ("ori", "dr", "rgb"))
    R = ori.shape[0]
    assert R > 31_000_000
    _, _, (planes, inv) = ops.ray_keys(ori, dr, rgb, full["w"], want_key=False, want_planes=True)
    small, large = ops.score_topk_workspace_bytes(R, 2, 100, planes=True), ops.score_topk_workspace_bytes(R, 2, 100)
    assert small < 0.8 * large                                     # 24-bit logits: 784 instead of 1024 B per ray and image
    ws = torch.empty(large, dtype=torch.uint8, device="cuda")     # (the bf16x6 comparison below needs the fp32-logits size)
    idx, val, sc, st = ops.score_topk(q, n_tok, None, 100, want_stats=True, workspace=ws[:small], key_planes=planes, key_scale=inv)
    # softmax mass: every token row sums to 1 over the rays -> the scores of an image sum to its token count
    tot = sc.double().sum(dim=1).cpu().numpy()
    assert np.allclose(tot, n_tok.cpu().numpy(), rtol=2e-4)
    assert bool(torch.isfinite(sc).all()) and float(sc.min()) >= 0.0
    mx, se = st[..., 0], st[..., 1]
    for b, t in enumerate(n_tok.tolist()):
        assert bool(torch.isfinite(mx[b, :t]).all()) and float(se[b, :t].min()) >= 1.0      # the max itself contributes exp(0)
    # top-k: sorted, unique, equal to the score vector at those indices, nothing outside beats the 100th
    for b in range(2):
        v, i = val[b], idx[b]
        assert bool((v[:-1] >= v[1:]).all()) and len(set(i.tolist())) == 100
        assert torch.equal(sc[b, i], v)
        rest = sc[b].clone()
        rest[i] = -1.0
        assert float(rest.max()) <= float(v[-1])
        ties = (sc[b] == v[-1]).nonzero().flatten()                # ties at the boundary resolve to the lowest indices
        assert int(i[v == v[-1]].max()) <= int(ties[int((v == v[-1]).sum()) - 1])
    # batch invariance: image 1 alone gives the same bits as inside the batch
    i1, v1, s1, _ = ops.score_topk(q[1:2].contiguous(), n_tok[1:2].contiguous(), None, 100, workspace=ws, key_planes=planes, key_scale=inv)
    assert torch.equal(i1[0], idx[1]) and torch.equal(s1[0], sc[1])
    # an independent arithmetic scheme (fp32 keys from the bf16 x 6 dense layers, 3 bf16 planes x 6 terms split on the fly, different kernel, different
    # logits layout) picks the same rays
    del s1
    ops.set_mma_mode(ops.MMA_BF16X6)
    try:
        _, key6 = ops.ray_keys(ori, dr, rgb, full["w"])
        i6, v6, s6, _ = ops.score_topk(q, n_tok, key6, 100, workspace=ws)
        del key6
    finally:
        ops.set_mma_mode(ops.MMA_DEFAULT)
    assert float((s6 - sc).abs().max() / sc.abs().max()) < 1e-5
    for b in range(2):
        assert set(i6[b].tolist()) == set(idx[b].tolist())
