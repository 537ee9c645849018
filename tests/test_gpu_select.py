"""The select path (sixdgs_score_select: top-k without the [T, R] logits ever reaching HBM) against the two-pass scorer and
the CPU oracle: same 100 rays in the same order wherever the gaps are real, values within fp32 rounding; plus the cases the
bounds cannot decide (too many near-ties for max_candidates, exponent overflow), which must be reported (status -1) and fall
back to the two-pass scorer inside IdentificationModule.score_tokens."""
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    o = importlib.import_module("6dgs_amd.ops")
    o.set_mma_mode(o.MMA_DEFAULT)
    return o


def make_case(ops, r, seed, q_scale, n_tok, key_spread=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    key = torch.randn(r, 384, generator=g) * 0.07
    if key_spread != 1.0:                       # groups of 128 rays with very different magnitudes: exercises the per-tile scales
        key = key * torch.exp(torch.randn(r // 128 + 1, 1, generator=g) * key_spread).repeat_interleave(128, 0)[:r]
    q = torch.randn(len(n_tok), 256, 384, generator=g) * q_scale
    nt = torch.tensor(n_tok, dtype=torch.int32)
    for b, t in enumerate(n_tok):
        q[b, t:] = 0.0
    key, q, nt = key.cuda(), q.cuda(), nt.cuda()
    planes, scale = ops.split_planes_f16(key)
    si = ops.select_sample_indices(r, "cuda")
    s_planes, s_scale = ops.split_planes_f16(key[si].contiguous())
    return dict(key=key, q=q, nt=nt, planes=planes, scale=scale, s_planes=s_planes, s_scale=s_scale, n_tok=list(n_tok))


def check_against_two_pass_and_oracle(ops, oracle, c, k=100, cmax=4096, oracle_images=(0,), allow_giveup=False, tol2=2e-6, tolo=1e-5):
    idx, val, status = ops.score_select(c["q"], c["nt"], c["planes"], c["scale"], c["s_planes"], c["s_scale"], k, max_candidates=cmax,
                                        n_tok_host=c["n_tok"])                      # host token counts: the images are packed into the sweep's tiles
    i2, v2, sc, _ = ops.score_topk(c["q"], c["nt"], None, k, key_planes=c["planes"], key_scale=c["scale"], want_scores=True)
    st = status.tolist()
    for b, t in enumerate(c["n_tok"]):
        if st[b] < 0 and allow_giveup:
            assert int(idx[b].max()) == -1        # no partial answers
            continue
        assert st[b] >= 0, f"image {b}: select path gave up ({st[b]})"
        if t == 0:
            assert torch.equal(idx[b], i2[b]) and float(val[b].abs().max()) == 0.0
            continue
        smax = float(v2[b][0])
        assert set(idx[b].tolist()) <= set(torch.nonzero(sc[b] >= v2[b][-1] - 2 * tol2 * smax).flatten().tolist())   # nothing clearly outside
        assert float((val[b] - sc[b][idx[b]]).abs().max()) / smax < tol2                                        # exact scores of those rays
        gaps = (v2[b][:-1] - v2[b][1:]) / smax
        if float(gaps.min()) > 2 * tol2:         # all gaps real: the same rays in the same order
            assert torch.equal(idx[b], i2[b])
        assert bool((val[b][:-1] >= val[b][1:]).all()) and len(set(idx[b].tolist())) == k
    key_np = c["key"].cpu().numpy()
    for b in oracle_images:
        t = c["n_tok"][b]
        if st[b] < 0:
            continue
        s_ref = oracle.attention_scores(c["q"][b, :t].cpu().numpy(), key_np)
        order = np.argsort(-s_ref, kind="stable")
        margin = 0.8 * tolo * float(s_ref.max())
        got = idx[b].cpu().numpy()
        must = order[:k][s_ref[order[:k]] - s_ref[order[k]] > margin]
        assert set(must.tolist()) <= set(got.tolist())
        assert float(s_ref[got].min()) >= float(s_ref[order[k - 1]]) - margin
        assert np.abs(val[b].cpu().numpy() - s_ref[got]).max() / float(s_ref.max()) < tolo
    return st


@pytest.mark.parametrize("q_scale,name", [(0.02, "flat"), (6.0, "moderate"), (45.0, "peaked"), (130.0, "very peaked")])
def test_select_matches_two_pass_and_oracle(ops, oracle, q_scale, name):
    """Four softmax regimes (logit spread over the rays = 0.07 q_scale: 0.0014 / 0.42 / 3.2 / 9.1), ragged token counts, R not a
    multiple of 256.  In the last one a handful of rays carry each token's whole softmax mass, the 1/16 sample misses most of
    them and the bounds may span more than max_candidates rays: giving up (status -1) is allowed there, a wrong answer is not."""
    c = make_case(ops, 1_200_037, 11, q_scale, (256, 137, 1, 200))
    # Tolerances.  A logit is a 384-term fp32 dot product: its rounding error is ~1e-7 * sum|q_k||key_k| / sqrt(384), i.e. ~1e-7 * 1.4
    # * q_scale in absolute terms, whatever the order of summation (matrix-core chain, scalar FMA chain of the re-score, the oracle's
    # loop).  A score dominated by one large logit inherits that error RELATIVELY (d e^x = e^x dx): 6e-6 at q_scale 45, 2e-5 at 130.
    # So two correct fp32 evaluations agree to 2e-6 / 1e-5 (vs two-pass / vs oracle, the bar of the parity tests) only while the logits
    # are small; beyond that the comparison is held to the logit error bound.
    tol2, tolo = (2e-6, 1e-5) if q_scale <= 6.0 else ((3e-5, 3e-5) if q_scale <= 45.0 else (1e-4, 1e-4))
    st = check_against_two_pass_and_oracle(ops, oracle, c, oracle_images=(0, 1), allow_giveup=name == "very peaked", tol2=tol2, tolo=tolo)
    print(f"[select {name}] candidates per image: {st}")
    assert max(st) <= 4096


def test_select_zero_token_image_and_tile_scales(ops, oracle):
    # 128-ray tiles whose magnitudes differ by up to ~20x.  All of the top rays then sit in a handful of tiles: the k-th largest TILE maximum of U
    # (the round-3 threshold) is far below the k-th largest U, more than max_candidates rays pass it, and the image must be caught by the per-image
    # exact selection inside sixdgs_select_candidates -- not refused (this very case was, until that fallback existed)
    c = make_case(ops, 1_100_000, 5, 1.0, (256, 0, 64), key_spread=0.7)
    st = check_against_two_pass_and_oracle(ops, oracle, c, oracle_images=(0, 2))
    assert st[1] == 0                                                              # the image without tokens: all scores are exactly 0
    # tiles 10^4 apart: the logits of the largest tiles run into the hundreds and the sample maximum is exceeded by more than
    # e^88 somewhere -- the select path must say so (and never answer wrongly)
    c = make_case(ops, 1_100_000, 5, 1.0, (256, 0, 64), key_spread=3.0)
    check_against_two_pass_and_oracle(ops, oracle, c, oracle_images=(), allow_giveup=True, tol2=1e-3)


def test_token_packing_is_invisible_and_matches_the_oracle(ops, oracle):
    """Round 5: views of 40 / 64 / 65 / 128 / 200 tokens (and 256, 137, none) in ONE batch.  With the host token counts the library packs them into
    256-token tiles (csrc/sweep_plan.h: 17 quarters of 64 tokens laid one after the other into 5 tiles, an image may span two); without them every image sweeps a tile of its own.  Same rays, same values,
    same candidate counts, bit for bit -- an image's U, per-token sums and sample statistics do not depend on the quarter it sits in nor on its neighbours --
    and both agree with the two-pass scorer and the CPU oracle."""
    n_tok = (40, 64, 65, 128, 200, 0, 256, 137)
    c = make_case(ops, 1_200_037, 31, 6.0, n_tok)
    plan = ops.select_sweep_plan(list(n_tok))
    assert plan == [(5, 8)], plan                                    # 5 tiles for 8 images (17 quarters; the image without tokens rides along)
    assert ops.select_sweep_plan(None, batch=8) == [(8, 8)]
    packed = ops.score_select(c["q"], c["nt"], c["planes"], c["scale"], c["s_planes"], c["s_scale"], 100, n_tok_host=list(n_tok))
    alone = ops.score_select(c["q"], c["nt"], c["planes"], c["scale"], c["s_planes"], c["s_scale"], 100)
    for a, b in zip(packed, alone):
        assert torch.equal(a, b)
    # every image by itself (a batch of one: nothing to pack with)
    for i in (0, 2, 4):
        one = ops.score_select(c["q"][i:i + 1].contiguous(), c["nt"][i:i + 1].contiguous(), c["planes"], c["scale"], c["s_planes"], c["s_scale"], 100,
                               n_tok_host=[n_tok[i]])
        assert torch.equal(one[0][0], packed[0][i]) and torch.equal(one[1][0], packed[1][i]) and int(one[2][0]) == int(packed[2][i])
    st = check_against_two_pass_and_oracle(ops, oracle, c, oracle_images=(0, 1, 2, 3, 4))
    assert st[5] == 0 and min(v for i, v in enumerate(st) if i != 5) >= 100


def test_token_packing_stage_by_stage_and_a_wrong_host_copy(ops):
    """The staged entry points (what the streamed and the ray-sharded scorers call) pack as well: U and g_t of a packed sweep over ragged chunks equal the
    unpacked ones bit for bit.  And a host copy that promises FEWER tokens than the device count is caught: that image comes back undecidable (status
    -1, scored by the two-pass path), never scored without some of its tokens."""
    n_tok = (64, 128, 60, 100, 256, 1)
    c = make_case(ops, 700_123, 41, 6.0, n_tok)
    runs = []
    for host in (list(n_tok), None):
        ss = ops.SelectStream(c["q"], c["nt"], 700_123, 100, 4096, host)
        ss.begin(c["s_planes"], c["s_scale"])
        for r0, r1 in ((0, 300_032), (300_032, 700_123)):
            planes, scale = ops.split_planes_f16(c["key"][r0:r1].contiguous())
            ss.sweep(planes, scale, r0)
        cand, count = ss.candidates()
        runs.append((ss.ctok.clone(), ss.gsum.clone(), ss.u[:, :700_123].clone(), count.clone()))
    for a, b in zip(*runs):
        assert torch.equal(a, b)
    wrong = list(n_tok)
    wrong[3] = 64                                                    # image 3 has 100 tokens; the host copy says 64 -> one quarter too few
    idx, val, status = ops.score_select(c["q"], c["nt"], c["planes"], c["scale"], c["s_planes"], c["s_scale"], 100, n_tok_host=wrong)
    good = ops.score_select(c["q"], c["nt"], c["planes"], c["scale"], c["s_planes"], c["s_scale"], 100, n_tok_host=list(n_tok))
    assert int(status[3]) == -1 and int(idx[3].max()) == -1
    for i in (0, 1, 2, 4, 5):
        assert torch.equal(idx[i], good[0][i]) and torch.equal(val[i], good[1][i])
    more = list(n_tok)
    more[0] = 256                                                    # promising MORE tokens than there are only wastes a tile: same answer
    idx2, val2, st2 = ops.score_select(c["q"], c["nt"], c["planes"], c["scale"], c["s_planes"], c["s_scale"], 100, n_tok_host=more)
    assert torch.equal(idx2, good[0]) and torch.equal(val2, good[1]) and torch.equal(st2, good[2])


def test_select_reports_what_it_cannot_decide(ops):
    """(a) a nearly flat softmax with room for 104 candidates only: more rays than that are within the bound's reach -> status -1;
    (b) a ray outside the sample whose logits exceed the sample maximum by far more than 88: e' overflows -> status -1."""
    c = make_case(ops, 1_200_000, 3, 0.002, (256, 256))
    _, _, status = ops.score_select(c["q"], c["nt"], c["planes"], c["scale"], c["s_planes"], c["s_scale"], 100, max_candidates=104)
    assert status.tolist() == [-1, -1]
    c = make_case(ops, 1_200_000, 4, 1.0, (256,))
    si = set(ops.select_sample_indices(1_200_000, "cpu").tolist())
    hot = next(i for i in range(500_000, 500_100) if i not in si)
    key = c["key"].clone()
    key[hot] = c["q"][0, 0] * 4000.0 / float(c["q"][0, 0].norm())        # logit of token 0 ~ 4000 * |q| / sqrt(384) >> 88
    planes, scale = ops.split_planes_f16(key)
    _, _, status = ops.score_select(c["q"], c["nt"], planes, scale, c["s_planes"], c["s_scale"], 100)
    assert status.tolist() == [-1]
    i2, v2, _, _ = ops.score_topk(c["q"], c["nt"], None, 100, key_planes=planes, key_scale=scale)
    assert int(i2[0, 0]) == hot                      # the two-pass scorer (online maximum) handles it


def test_module_takes_the_select_path_and_falls_back(ops, syn, oracle):
    """IdentificationModule.score_tokens(want_scores=False) on a scene above SELECT_MIN_RAYS: select path, same answer as with
    want_scores=True (two-pass); with max_candidates forced tiny every image falls back and the answer is still the same."""
    pkg = importlib.import_module("6dgs_amd")
    idm = pkg.IdentificationModule("dino")
    idm.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_scorer_state_dict(0).items()}, strict=False)
    idm = idm.cuda().eval()
    rays = syn.make_rays(1_300_000, 2)
    o, d, c = (torch.from_numpy(rays[k]).cuda() for k in ("ori", "dir", "rgb"))
    toks = [torch.from_numpy(syn.make_tokens(t, 30 + i, 40.0)).cuda() for i, t in enumerate((256, 173, 256))]
    i_s, v_s, none = idm.score_tokens(toks, o, d, c, 100, want_scores=False)
    assert none is None and idm.last_scoring_path == "select" and min(idm.last_select_candidates) >= 100
    i_t, v_t, sc = idm.score_tokens(toks, o, d, c, 100, want_scores=True)
    assert idm.last_scoring_path == "two-pass"
    for b in range(3):
        assert set(i_s[b].tolist()) == set(i_t[b].tolist())
        assert float((v_s[b] - sc[b][i_s[b]]).abs().max() / v_t[b][0]) < 3e-5      # logits of magnitude ~10: fp32 dot-product rounding x e^x
    flat = [t * 0.0005 for t in toks]           # nearly flat softmax: far more than 104 rays within the bound's reach
    i_ft, v_ft, sc_f = idm.score_tokens(flat, o, d, c, 100, want_scores=True)
    old = ops.SELECT_MAX_CANDIDATES
    try:
        ops.SELECT_MAX_CANDIDATES = 104
        i_f, v_f, _ = idm.score_tokens(flat, o, d, c, 100, want_scores=False)
        assert idm.last_scoring_path == "select+two-pass(3)" and idm.last_select_candidates == [-1, -1, -1]
    finally:
        ops.SELECT_MAX_CANDIDATES = old
    assert torch.equal(i_f, i_ft) and torch.equal(v_f, v_ft)       # the fallback IS the two-pass scorer
    ops.set_select_enabled(False)
    try:
        i_n, v_n, _ = idm.score_tokens(toks, o, d, c, 100, want_scores=False)
        assert idm.last_scoring_path == "two-pass" and torch.equal(i_n, i_t) and torch.equal(v_n, v_t)
    finally:
        ops.set_select_enabled(True)


def test_select_stage_by_stage_over_chunks_matches_the_resident_call(ops):
    """sixdgs_select_begin / _sweep (3 ragged chunks) / _candidates / _rescore on COMPACT planes of the candidates alone -- what the
    streamed scorer runs for scenes whose key planes exceed the GPU -- against the one-call resident path."""
    c = make_case(ops, 1_200_037, 21, 6.0, (256, 137, 200))
    idx, val, status = ops.score_select(c["q"], c["nt"], c["planes"], c["scale"], c["s_planes"], c["s_scale"], 100)
    assert min(status.tolist()) >= 100
    ss = ops.SelectStream(c["q"], c["nt"], 1_200_037, 100, 4096, c["n_tok"])
    ss.begin(c["s_planes"], c["s_scale"])
    for r0, r1 in ((0, 500_224), (500_224, 900_096), (900_096, 1_200_037)):
        planes, scale = ops.split_planes_f16(c["key"][r0:r1].contiguous())
        ss.sweep(planes, scale, r0)
    cand, count = ss.candidates()
    assert count.tolist() == status.tolist()                       # the same candidate sets (U is the same bits, g to rounding)
    inside = torch.arange(4096, device="cuda")[None] < count[:, None]
    ci = torch.where(inside, cand, torch.zeros_like(cand)).reshape(-1)
    cp, cs = ops.split_planes_f16(c["key"][ci].contiguous())
    i2, v2, st2 = ss.rescore(cp, cs, cand, count, compact=True)
    assert st2.tolist() == status.tolist()
    for b in range(3):
        assert set(i2[b].tolist()) == set(idx[b].tolist())
        assert float((v2[b] - val[b]).abs().max() / val[b][0]) < 2e-6
    # and on the scene's own planes (compact = False) the staged path is the resident call
    i3, v3, st3 = ss.rescore(c["planes"], c["scale"], cand, count, compact=False)
    for b in range(3):
        assert torch.equal(i3[b], idx[b]) and float((v3[b] - val[b]).abs().max() / val[b][0]) < 1e-6


def _exact_scores(q, nt, key):
    """fp64 softmax-over-rays column sums of the fp32 operands (the quantity every scorer approximates), on the GPU."""
    out = []
    for b in range(q.shape[0]):
        t = int(nt[b])
        logits = (q[b, :t].double() @ key.double().T) / (384.0 ** 0.5)
        out.append(torch.softmax(logits, dim=-1).sum(0))
    return out


def test_key_norm_max_bounds_every_row_norm_tightly(ops):
    c = make_case(ops, 300_001, 8, 1.0, (256,), key_spread=0.7)
    n = ops.key_norm_max(c["planes"], c["scale"])
    true = float(c["key"].double().norm(dim=1).max())
    assert true <= float(n) <= true * 1.0005                                       # an upper bound, within the stated rounding-up
    acc = torch.zeros(1, device="cuda")                                              # chunk by chunk into one scalar
    for r0 in range(0, 300_001, 100_096):
        p, s = ops.split_planes_f16(c["key"][r0:r0 + 100_096].contiguous())
        ops.key_norm_max(p, s, out=acc)
    assert abs(float(acc) - float(n)) <= 1e-6 * float(n)


def test_key_norm_from_the_k_proj_epilogue_equals_the_pass_over_the_planes(ops, syn):
    """sixdgs_ray_keys_ex(d_key_norm_max): when k_proj writes the key planes itself the norms come out of its epilogue (no extra pass
    over 1536 B per ray -- 0.2 s per step of the streamed cfg-4 scorer otherwise); same bound as sixdgs_key_planes_norm_max of the
    finished planes, accumulated chunk by chunk, ragged last tile included."""
    rays = syn.make_rays(300_037, 9)
    o, d, c = (torch.from_numpy(rays[k]).cuda() for k in ("ori", "dir", "rgb"))
    w = ops.PackedWeights({k: torch.from_numpy(v) for k, v in syn.make_scorer_state_dict(0).items()}, "cuda")
    norm = torch.zeros(1, device="cuda")
    _, _, (planes, scale) = ops.ray_keys(o, d, c, w, want_key=False, want_planes=True, norm_out=norm, max_chunk=65536)
    ref = ops.key_norm_max(planes, scale)
    _, key = ops.ray_keys(o, d, c, w)
    true = float(key.double().norm(dim=1).max())
    assert true <= float(norm) <= true * 1.0006 and abs(float(norm) - float(ref)) <= 3e-6 * float(ref)
    again = torch.zeros(1, device="cuda")
    ops.ray_keys(o, d, c, w, want_key=False, want_planes=True, norm_out=again, max_chunk=65536)
    assert torch.equal(again, norm)                                                  # deterministic
    # and the keys themselves against the oracle at a size that spans several chunks of the chain and ends in a ragged tile (the
    # ray MLP was oracle-checked at R <= 4096 only): 300 037 rays, every 7th compared (the OpenMP oracle takes seconds for those)
    from oracle import oracle as O
    O.build()
    pick = np.arange(0, 300_037, 7)
    _, k_ref = O.ray_features(rays["ori"][pick], rays["dir"][pick], rays["rgb"][pick], syn.make_scorer_state_dict(0))
    k_hip = key[torch.from_numpy(pick).cuda()].cpu().numpy()
    assert np.abs(k_hip - k_ref).max() / np.abs(k_ref).max() < 5e-6
    both = torch.zeros(1, device="cuda")                                             # the non-fused path (fp32 keys wanted too) fills it as well
    ops.ray_keys(o, d, c, w, want_key=True, want_planes=True, norm_out=both)
    assert abs(float(both) - float(ref)) <= 3e-6 * float(ref)


def test_select_slack_follows_the_logit_error_bound_single_token_large_logits(ops):
    """VERDICT r2 #9 / ADVICE r2: with ONE token g_min = g_max and the derived slack (1 - eps) / (1 + eps), eps = 1.4e-4 x + 1.3e-5,
    x = |q| max|k| / sqrt(384), is all that separates the candidates from the rest.  Logits up to ~80 (x ~ 300): the sweep's U is
    then off the exact value by up to ~1e-3 relative -- far beyond the old constant 2^-16 -- and the answer must still be the exact
    top-100 (or a refusal, never a wrong set)."""
    r = 1_200_000
    g = torch.Generator(device="cpu").manual_seed(77)
    key = (torch.randn(r, 384, generator=g) * 0.07).cuda()
    q = torch.zeros(2, 256, 384)
    q[:, 0] = torch.randn(2, 384, generator=g) * torch.tensor([[230.0], [120.0]])
    q = q.cuda()
    nt = torch.tensor([1, 1], dtype=torch.int32).cuda()
    planes, scale = ops.split_planes_f16(key)
    si = ops.select_sample_indices(r, "cuda")
    sp, ss = ops.split_planes_f16(key[si].contiguous())
    idx, val, status = ops.score_select(q, nt, planes, scale, sp, ss, 100)
    exact = _exact_scores(q, nt, key)
    for b in range(2):
        lmax = float(((q[b, 0].double() @ key.double().T) / 384.0 ** 0.5).max())
        assert lmax > (60.0 if b == 0 else 30.0)
        st = int(status[b])
        if st < 0:
            assert int(idx[b].max()) == -1
            continue
        s = exact[b]
        order = torch.argsort(s, descending=True, stable=True)
        # an fp32-operand logit carries ~1e-7 x of rounding however it is summed: two rays closer than that (relative, as e^dx) are a tie
        x = float(q[b, 0].double().norm() * key.double().norm(dim=1).max() / 384.0 ** 0.5)
        tie = 4e-7 * x
        must = order[:100][(s[order[:100]] / s[order[100]] - 1.0) > tie]
        got = set(idx[b].tolist())
        assert set(must.tolist()) <= got and len(got) == 100
        assert float(s[idx[b]].min()) >= float(s[order[99]]) * (1.0 - tie)
        assert float(((val[b].double() - s[idx[b]]).abs() / s[idx[b]]).max()) < max(2e-6, 2e-7 * x)
        assert st >= 100


def test_select_near_ties_at_the_threshold_with_large_logits(ops):
    """Rays planted a hair below and above the 100th score (copies of the rays ranked 90..110 whose keys are scaled by 1 +- j 2e-7,
    i.e. logits moved by ~1e-5 at |logit| ~ 30) under large logits: every ray whose exact score clears the 100th by more than the fp32
    logit noise must be returned, nothing below it by more than that may be; the candidate set has to contain all of them."""
    r = 1_200_000
    c = make_case(ops, r, 31, 45.0, (256, 64))
    key = c["key"].clone()
    s0 = _exact_scores(c["q"], c["nt"], key)[0]
    order = torch.argsort(s0, descending=True)
    src = order[90:110]
    dst = torch.arange(700_000, 700_000 + 400, device="cuda")
    dst = dst[~torch.isin(dst, order[:200])]
    fac = 1.0 + (torch.arange(dst.numel(), device="cuda").float() - dst.numel() / 2) * 2e-7
    key[dst] = key[src[torch.arange(dst.numel(), device="cuda") % 20]] * fac[:, None]
    planes, scale = ops.split_planes_f16(key)
    si = ops.select_sample_indices(r, "cuda")
    sp, ss = ops.split_planes_f16(key[si].contiguous())
    idx, val, status = ops.score_select(c["q"], c["nt"], planes, scale, sp, ss, 100)
    i2, v2, sc, _ = ops.score_topk(c["q"], c["nt"], None, 100, key_planes=planes, key_scale=scale, want_scores=True)
    exact = _exact_scores(c["q"], c["nt"], key)
    for b in range(2):
        assert int(status[b]) >= 100, status.tolist()
        s = exact[b]
        o = torch.argsort(s, descending=True, stable=True)
        tie = 3e-5                                   # q_scale 45: fp32 dot-product rounding x e^x (the tolerance of the 'peaked' case above)
        must = o[:100][(s[o[:100]] / s[o[100]] - 1.0) > tie]
        got = set(idx[b].tolist())
        assert set(must.tolist()) <= got and len(got) == 100
        assert float(s[idx[b]].min()) >= float(s[o[99]]) * (1.0 - tie)
        assert float(((val[b].double() - s[idx[b]]).abs() / s[o[0]]).max()) < tie
        assert float((val[b] - sc[b][idx[b]]).abs().max() / v2[b][0]) < tie          # and the two-pass scorer agrees on those rays


def test_deferred_status_whole_step_graph_and_resolve(ops, syn):
    """the select path's one host read (B status ints) deferred to the step's own D2H, so that image side + q_proj +
    select + pose solve capture into ONE hipGraph.  (a) deferred == immediate; (b) a captured step replayed on NEW images gives the
    eager poses of those images; (c) images the select path refuses (forced: room for 104 candidates under a flat softmax) are re-done
    by resolve_poses with the two-pass scorer's answer."""
    pkg = importlib.import_module("6dgs_amd")
    tp = importlib.import_module("6dgs_amd.test")
    idm = pkg.IdentificationModule("dino")
    idm.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_scorer_state_dict(0, with_cnn=True).items()}, strict=False)
    idm = idm.cuda().eval()
    rays = syn.make_rays(1_200_000, 3)
    o, d, c = (torch.from_numpy(rays[k]).cuda() for k in ("ori", "dir", "rgb"))

    def batch(seed):
        return [torch.from_numpy(np.ascontiguousarray(cam["image"])).cuda() for cam in syn.make_cameras(2, seed, width=160, height=120)]

    imgs = batch(50)
    eager = tp.estimate_poses(idm, imgs, o, d, c)
    assert idm.last_scoring_path == "select"
    sol = tp.estimate_poses(idm, imgs, o, d, c, defer_status=True)
    assert sol["packed"].shape == (2, 17) and sol["pending_select"] is not None
    c2w = tp.resolve_poses(idm, sol, sol["packed"].cpu())
    assert torch.equal(c2w, eager["c2w"].cpu()) and torch.equal(sol["idx"], eager["idx"]) and min(idm.last_select_candidates) >= 100
    # (b) one hipGraph for the whole step
    static = [im.clone() for im in imgs]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        tp.estimate_poses(idm, static, o, d, c, defer_status=True, image_graph=False)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        gsol = tp.estimate_poses(idm, static, o, d, c, defer_status=True, image_graph=False)
    for seed in (51, 52):
        new = batch(seed)
        for s_, n_ in zip(static, new):
            s_.copy_(n_)
        g.replay()
        got = tp.resolve_poses(idm, gsol, gsol["packed"].cpu())
        want = tp.estimate_poses(idm, new, o, d, c, image_graph=False)
        assert torch.equal(gsol["idx"], want["idx"]), seed
        assert float((got - want["c2w"].cpu()).abs().max()) < 1e-6
    # (c) refused images
    flat = [t * 0.0005 for t in (torch.from_numpy(syn.make_tokens(256, 60 + i, 40.0)).cuda() for i in range(2))]
    up = torch.nn.functional.normalize(torch.tensor([[0.1, 0.9, 0.2], [0.3, -0.5, 0.8]], device="cuda"), dim=-1)
    ref = tp.estimate_poses(idm, None, o, d, c, tokens=flat, up=up, want_scores=True)          # two-pass scorer
    old = ops.SELECT_MAX_CANDIDATES
    try:
        ops.SELECT_MAX_CANDIDATES = 104
        idm._select_ws = None
        sol = tp.estimate_poses(idm, None, o, d, c, tokens=flat, up=up, defer_status=True)
        host = sol["packed"].cpu()
        assert host[:, 16].tolist() == [-1.0, -1.0]
        c2w = tp.resolve_poses(idm, sol, host)
    finally:
        ops.SELECT_MAX_CANDIDATES = old
        idm._select_ws = None
    assert idm.last_scoring_path == "select+two-pass(2)"
    assert torch.equal(sol["idx"], ref["idx"]) and float((c2w - ref["c2w"].cpu()).abs().max()) < 1e-6


_CHILD = r"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ["SIXDGS_TEST_ROOT"]); sys.path.insert(0, os.path.join(os.environ["SIXDGS_TEST_ROOT"], "tests"))
import test_gpu_select as t
ops = importlib.import_module("6dgs_amd.ops"); ops.set_mma_mode(ops.MMA_DEFAULT)
c = t.make_case(ops, 300_000, 77, 6.0, [256, 200, 256, 31])
out = None
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    out = ops.score_select(c["q"], c["nt"], c["planes"], c["scale"], c["s_planes"], c["s_scale"], 100, max_candidates=4096)
    torch.cuda.synchronize(); dt = time.time() - t0
np.savez(sys.argv[1], idx=out[0].cpu().numpy(), val=out[1].cpu().numpy(), status=out[2].cpu().numpy(), seconds=dt)
"""


def test_sibling_meeting_gives_up_on_a_member_that_never_arrives(ops, tmp_path):
    """The per-tile meeting of a persistent sibling set is a cache-locality measure with a BOUNDED wait (score.hip, kSibSpinLimit): with
    SIXDGS_SIBLING_SYNC=3 every set waits for one arrival more than it has members -- what a second sweep holding the CUs of half a set looks
    like -- and must release itself after the limit and return the same result, in bounded time."""
    import subprocess
    outs = {}
    for mode in ("3", "0", "2"):
        f = str(tmp_path / f"m{mode}.npz")
        env = dict(os.environ, SIXDGS_SIBLING_SYNC=mode, SIXDGS_TEST_ROOT=ROOT)
        p = subprocess.run([sys.executable, "-W", "ignore", "-c", _CHILD, f], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs[mode] = np.load(f)
    a, b = outs["3"], outs["0"]
    assert np.array_equal(outs["2"]["idx"], b["idx"]) and np.array_equal(outs["2"]["val"], b["val"]) and np.array_equal(outs["2"]["status"], b["status"])      # persistent sets without the meeting: the same bits as the one-shot grid
    assert (a["status"] >= 0).all() and np.array_equal(a["status"] >= 0, b["status"] >= 0)
    assert np.array_equal(a["idx"], b["idx"]) and np.allclose(a["val"], b["val"], rtol=1e-6, atol=0)
    assert float(a["seconds"]) < 2.0, float(a["seconds"])          # 64 sets give up once each, concurrently: tens of milliseconds
    c = make_case(ops, 300_000, 77, 6.0, [256, 200, 256, 31])      # and the default layout (lock-step, everyone arrives) in this process
    idx, val, st = ops.score_select(c["q"], c["nt"], c["planes"], c["scale"], c["s_planes"], c["s_scale"], 100, max_candidates=4096)
    assert np.array_equal(idx.cpu().numpy(), b["idx"])


def test_sweep_launch_grouping_is_invisible(ops, tmp_path):
    """Round 4: a batch of more than 12 images is swept in launches of 8 (the images of a launch share an XCD's L2; `sixdgs_select_sweep`).  The grouping must not
    show: 14 images (launches of 8 + 6; ragged token counts) give the same idx / val / status bit for bit as ONE launch of 14 (SIXDGS_SWEEP_MAX_IMAGES=0) and as
    launches of 4 -- and each image the same as alone."""
    import subprocess
    child = _CHILD.replace("[256, 200, 256, 31]", "[256, 200, 256, 31, 256, 137, 256, 256, 64, 256, 1, 256, 190, 256]")
    outs = {}
    for cap in ("", "0", "4"):
        f = str(tmp_path / f"cap{cap or 'default'}.npz")
        env = dict(os.environ, SIXDGS_TEST_ROOT=ROOT)
        env.pop("SIXDGS_SWEEP_MAX_IMAGES", None)
        if cap:
            env["SIXDGS_SWEEP_MAX_IMAGES"] = cap
        p = subprocess.run([sys.executable, "-W", "ignore", "-c", child, f], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs[cap] = np.load(f)
    a = outs[""]
    assert a["idx"].shape == (14, 100) and (a["status"][[0, 1, 2, 4]] >= 100).all()
    for cap in ("0", "4"):
        b = outs[cap]
        assert np.array_equal(a["idx"], b["idx"]) and np.array_equal(a["status"], b["status"]), cap
        assert np.array_equal(np.nan_to_num(a["val"], nan=-7.0), np.nan_to_num(b["val"], nan=-7.0)), cap
    c = make_case(ops, 300_000, 77, 6.0, [256, 200, 256, 31, 256, 137, 256, 256, 64, 256, 1, 256, 190, 256])
    for i in (1, 9, 13):
        idx, val, st = ops.score_select(c["q"][i:i + 1].contiguous(), c["nt"][i:i + 1].contiguous(), c["planes"], c["scale"], c["s_planes"], c["s_scale"], 100, max_candidates=4096)
        assert np.array_equal(idx.cpu().numpy()[0], a["idx"][i]) and int(st[0]) == int(a["status"][i])


def test_prepass_terms_do_not_change_the_answer(ops, tmp_path):
    """Round 6: the sample pre-pass issues one MFMA term of three (its statistics set the sweep's exponent offsets and scale only; g_t is exact relative to
    them).  Against SIXDGS_PREPASS_TERMS=3: the same top-100 lists, values to 1e-6 (the exact re-score's, relative to slightly different offsets), about as many
    candidates -- over ragged token counts incl. a 1-token and a 31-token image."""
    import subprocess
    child = _CHILD.replace("[256, 200, 256, 31]", "[256, 200, 256, 31, 1, 137, 64, 256]")
    outs = {}
    for terms in ("", "3"):
        f = str(tmp_path / f"terms{terms or 'default'}.npz")
        env = dict(os.environ, SIXDGS_TEST_ROOT=ROOT)
        env.pop("SIXDGS_PREPASS_TERMS", None)
        if terms:
            env["SIXDGS_PREPASS_TERMS"] = terms
        p = subprocess.run([sys.executable, "-W", "ignore", "-c", child, f], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs[terms] = np.load(f)
    a, b = outs[""], outs["3"]
    assert a["idx"].shape == (8, 100) and np.array_equal(a["idx"], b["idx"])
    assert np.allclose(np.nan_to_num(a["val"], nan=-7.0), np.nan_to_num(b["val"], nan=-7.0), rtol=1e-6, atol=0)
    assert (np.sign(a["status"]) == np.sign(b["status"])).all() and (np.abs(a["status"] - b["status"]) <= np.maximum(8, b["status"] // 10)).all(), (a["status"], b["status"])


def test_streamed_scene_with_an_arena_survives_a_refused_image(ops, syn):
    """with an arena installed a scene is streamed BECAUSE its planes exceed the arena -- the two-pass fallback of the streamed select path
    (a refused image) used to carve every chunk's planes from the arena in both sweeps without giving them back and died with "arena exhausted".  Here:
    an arena that holds ~2.3 chunks of planes, a 1.3 M-ray scene streamed in chunks of 2^18 rays (6 chunks, two sweeps), every image refused (room for 104
    candidates under a flat softmax) -> the answer is the two-pass scorer's, the arena is back at its mark, and the same call works again."""
    pkg = importlib.import_module("6dgs_amd")
    idm = pkg.IdentificationModule("dino")
    idm.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_scorer_state_dict(0).items()}, strict=False)
    idm = idm.cuda().eval()
    rays = syn.make_rays(1_300_000, 2)
    o, d, c = (torch.from_numpy(rays[k]).cuda() for k in ("ori", "dir", "rgb"))
    flat = [torch.from_numpy(syn.make_tokens(t, 30 + i, 40.0)).cuda() * 0.0005 for i, t in enumerate((256, 173))]
    chunk = 1 << 18
    i_ref, v_ref = idm.score_tokens_streamed(flat, o, d, c, 100, chunk_rays=chunk, use_select=False)          # torch's allocator: the reference answer
    need_ws = ops.score_topk_workspace_bytes(chunk, 2, 100, planes=True)
    arena = ops.Arena(need_ws + int(2.3 * chunk * 1536) + ops.ray_keys_workspace_bytes(chunk, ops.RAY_KEYS_CHUNK_MIN) + (1300000 // 16 + 256) * 1536 + (64 << 20), "cuda")
    assert arena.capacity < 1_300_000 * 1536            # the scene's planes do NOT fit: that is why it is streamed
    old, prev = ops.SELECT_MAX_CANDIDATES, ops.set_arena(arena)
    try:
        ops.SELECT_MAX_CANDIDATES = 104
        for _ in range(2):
            i_a, v_a = idm.score_tokens_streamed(flat, o, d, c, 100, chunk_rays=chunk)
            assert idm.last_scoring_path == "streamed select+two-pass(2)"
            assert torch.equal(i_a, i_ref) and torch.equal(v_a, v_ref)
        held = arena.mark()                               # what stays taken between calls: the scene's ray sample, nothing per batch or per chunk
        i_b, v_b = idm.score_tokens_streamed(flat, o, d, c, 100, chunk_rays=chunk, use_select=False)
        assert arena.mark() == held and torch.equal(i_b, i_ref)
        assert arena.high <= arena.capacity
    finally:
        ops.SELECT_MAX_CANDIDATES = old
        ops.set_arena(prev)
        idm.invalidate_caches()
