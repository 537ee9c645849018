"""k_proj folded into ray-MLP layer 4 on the key-cache path (round 5): K = (Wk W4) h3 + (Wk b4 + bk).

Layer 4 has no non-linearity behind it (reference ray_preprocessor.py:27-31,46) and k_proj follows at once
(our_multihead_attention.py:74), so the plane-to-plane chain runs four layers instead of five.  The five-layer form stays
wherever features are returned (want_feat -> the fp32-operand kernels) and behind SIXDGS_FOLD_KPROJ=0.  Checked here:
the folded keys against (a) the five-layer plane chain, (b) the fp32-operand five-layer kernels, (c) the oracle's own unfused
MLP + k_proj; and that the top-100 of the four golden g5 regimes does not move."""
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from conftest import rel_err  # noqa: E402


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    o = importlib.import_module("6dgs_amd.ops")
    o.set_mma_mode(o.MMA_DEFAULT)
    return o


def row_err(a, b):
    """max over rows of (max |a - b| of the row) / (max |b| of the row)"""
    a, b = a.double(), b.double()
    return float(((a - b).abs().amax(dim=1) / b.abs().amax(dim=1).clamp_min(1e-30)).max())


@pytest.fixture(scope="module")
def keys(ops, oracle, syn):
    sd = syn.make_scorer_state_dict(0)
    rays = syn.make_rays(4096, 0)
    w = ops.PackedWeights({k: torch.from_numpy(v) for k, v in sd.items()}, "cuda")
    o, d, c = G(rays["ori"]), G(rays["dir"]), G(rays["rgb"])
    assert os.environ.get("SIXDGS_FOLD_KPROJ") is None
    _, k_fold, (p_fold, s_fold) = ops.ray_keys(o, d, c, w, want_planes=True)
    norm_fold = torch.zeros(1, device="cuda")
    _, _, (p_fold2, s_fold2) = ops.ray_keys(o, d, c, w, want_key=False, want_planes=True, norm_out=norm_fold)      # planes straight from the last layer
    os.environ["SIXDGS_FOLD_KPROJ"] = "0"
    try:
        _, k_five, (p_five, s_five) = ops.ray_keys(o, d, c, w, want_planes=True)
        norm_five = torch.zeros(1, device="cuda")
        ops.ray_keys(o, d, c, w, want_key=False, want_planes=True, norm_out=norm_five)
    finally:
        del os.environ["SIXDGS_FOLD_KPROJ"]
    f_ref, k_ref = ops.ray_keys(o, d, c, w, want_feat=True)                                # fp32-operand kernels: always five layers
    _, okey = oracle.ray_features(rays["ori"], rays["dir"], rays["rgb"], sd)
    torch.cuda.synchronize()
    return dict(sd=sd, w=w, k_fold=k_fold, p_fold=p_fold, s_fold=s_fold, p_fold2=p_fold2, s_fold2=s_fold2, k_five=k_five, p_five=p_five, s_five=s_five,
                k_ref=k_ref, okey=okey, norm_fold=norm_fold, norm_five=norm_five)


def test_folded_keys_row_wise(keys):
    assert not torch.equal(keys["k_fold"], keys["k_five"])          # the switch does switch
    assert row_err(keys["k_fold"], keys["k_five"]) <= 2e-6
    assert row_err(keys["k_fold"], keys["k_ref"]) <= 2e-6
    assert rel_err(keys["k_fold"].cpu().numpy(), keys["okey"]) < 5e-6
    assert row_err(keys["k_fold"], torch.from_numpy(keys["okey"]).cuda()) <= 5e-6
    # the folded last layer writes the scorer's planes itself: identical to one split pass over its fp32 keys
    assert torch.equal(keys["p_fold"], keys["p_fold2"]) and torch.equal(keys["s_fold"], keys["s_fold2"])
    # key-norm maximum of the scene (the select path's slack): the folded epilogue's against the rows themselves
    n_rows = float(keys["k_fold"].double().norm(dim=1).max())
    assert n_rows <= float(keys["norm_fold"]) <= n_rows * (1 + 1e-3)
    assert abs(float(keys["norm_fold"]) - float(keys["norm_five"])) <= 1e-5 * n_rows


@pytest.mark.parametrize("tag,T,scale", [("flat256", 256, 1.0), ("peaky256", 256, 40.0), ("peaky137", 137, 40.0), ("mid1", 1, 10.0)])
def test_folded_keys_same_top100(ops, syn, golden, keys, tag, T, scale):
    g = golden("g5_scorer")
    tok = syn.make_tokens(T, 1, scale)
    tokens, n_tok = ops.pad_tokens([G(tok)], "cuda")
    q = ops.q_proj(tokens, n_tok, keys["w"])
    i_f, v_f, s_f, _ = ops.score_topk(q, n_tok, None, 100, key_planes=keys["p_fold"], key_scale=keys["s_fold"], want_scores=True)
    i_5, v_5, s_5, _ = ops.score_topk(q, n_tok, None, 100, key_planes=keys["p_five"], key_scale=keys["s_five"], want_scores=True)
    assert rel_err(s_f.cpu().numpy(), s_5.cpu().numpy()) < 4e-6
    assert rel_err(s_f[0].cpu().numpy(), g[f"{tag}_scores"]) < 1e-5
    assert set(i_f[0].tolist()) == set(i_5[0].tolist()) == set(g[f"{tag}_idx"].tolist())
    if tag != "flat256":
        assert torch.equal(i_f, i_5) and (i_f[0].cpu().numpy() == g[f"{tag}_idx"]).all()
    # the select path on the folded planes: the same rays again
    si = ops.select_sample_indices(4096, "cuda")
    sp, ss = ops.split_planes_f16(keys["k_fold"][si].contiguous())
    i_s, v_s, st = ops.score_select(q, n_tok, keys["p_fold"], keys["s_fold"], sp, ss, 100)
    if int(st[0]) >= 0:
        assert set(i_s[0].tolist()) == set(i_f[0].tolist())
