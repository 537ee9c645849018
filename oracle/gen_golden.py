#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ by IMPORTING the reference.

TEST INFRASTRUCTURE ONLY.  Runs only in the build container, where /root/reference exists
(read-only).  Nothing from the reference is copied: this script imports its Python modules,
calls the functions on the hot path with seeded synthetic inputs from
``6dgs_amd/synthetic.py`` and stores inputs + outputs as small ``.npz`` fixtures.  The GPU box
only ever sees the fixtures.

Stubs: ``plyfile`` and ``simple_knn._C`` (imported at module load by
``scene/gaussian_model.py:23,25`` but never used on the pose path), ``torchvision.transforms``
and ``torch.hub.load`` (DINOv2 weights are not downloadable; the image tokens are an *input*
of the boundary).

Usage:  python oracle/gen_golden.py [--only g1,g5]
"""
from __future__ import annotations

import argparse
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("SIXDGS_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
syn = importlib.import_module("6dgs_amd.synthetic")


# --------------------------------------------------------------------------------------
# reference import with stubs
# --------------------------------------------------------------------------------------
def _install_stubs():
    ply = types.ModuleType("plyfile")
    ply.PlyData = object
    ply.PlyElement = object
    sys.modules["plyfile"] = ply
    sk = types.ModuleType("simple_knn")
    skc = types.ModuleType("simple_knn._C")
    skc.distCUDA2 = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub"))
    sys.modules["simple_knn"] = sk
    sys.modules["simple_knn._C"] = skc

    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")

    class _Identity:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x

    class _Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class _Interp:
        BICUBIC = "bicubic"
        BILINEAR = "bilinear"

    class _Resize:
        """Functional stand-in (torchvision is absent): plain F.interpolate to a square."""

        def __init__(self, size, interpolation="bilinear", antialias=True):
            self.size, self.mode = size, interpolation

        def __call__(self, x):
            return torch.nn.functional.interpolate(
                x, size=(self.size, self.size), mode=self.mode, align_corners=False, antialias=True)

    class _CenterCrop:
        def __init__(self, size):
            self.size = size

        def __call__(self, x):
            h, w = x.shape[-2:]
            t, l = (h - self.size) // 2, (w - self.size) // 2
            return x[..., t:t + self.size, l:l + self.size]

    class _Normalize:
        def __init__(self, mean, std):
            self.mean = torch.tensor(mean).view(1, 3, 1, 1)
            self.std = torch.tensor(std).view(1, 3, 1, 1)

        def __call__(self, x):
            return (x - self.mean) / self.std

    tr.Resize, tr.CenterCrop, tr.Normalize, tr.Compose = _Resize, _CenterCrop, _Normalize, _Compose
    tr.InterpolationMode = _Interp
    tv.transforms = tr
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tr


class FakeDino(torch.nn.Module):
    """Stand-in for dinov2_vits14: a fixed patch-embed (14x14 conv) so tokens depend on the
    image.  Only ``forward_features(x)['x_norm_patchtokens']`` is consumed (backbone.py:91-93)."""

    def __init__(self):
        super().__init__()
        g = np.random.default_rng(77)
        w = (g.standard_normal((384, 3, 14, 14)) / np.sqrt(3 * 14 * 14)).astype(np.float32)
        self.proj = torch.nn.Conv2d(3, 384, 14, stride=14, bias=False)
        with torch.no_grad():
            self.proj.weight.copy_(torch.from_numpy(w))

    def forward_features(self, x):
        y = self.proj(x)  # [1,384,16,16]
        return {"x_norm_patchtokens": y.flatten(2).transpose(1, 2)}


def import_reference():
    _install_stubs()
    sys.path.insert(0, REF)
    torch.hub.load = lambda *a, **k: FakeDino()
    mods = {}
    for name in (
        "pose_estimation.isocell", "pose_estimation.quadricell", "pose_estimation.ray_preprocessor",
        "pose_estimation.our_multihead_attention", "pose_estimation.line_intersection",
        "pose_estimation.sym_eig_3x3", "pose_estimation.error_computation",
        "pose_estimation.sampling", "pose_estimation.identification_module", "pose_estimation.test",
        "utils.sh_utils", "utils.general_utils", "scene.gaussian_model", "scene.scene_structure",
    ):
        mods[name.split(".")[-1]] = importlib.import_module(name)
    return mods


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def N(x):
    return x.detach().cpu().numpy()


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"  wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB, {len(arrs)} arrays)")


# --------------------------------------------------------------------------------------
# g1: a2 mask_degraded_ellipsoids, a6 compute_quadricell_centers, a7 mask_and_compute_rays
# --------------------------------------------------------------------------------------
def g1(m):
    q, gu = m["quadricell"], m["general_utils"]
    rng = np.random.default_rng(11)
    # a2: wide range of anisotropy so both outcomes occur
    sc = np.exp(rng.normal(-3.0, 1.6, size=(4096, 3))).astype(np.float32)
    mask = q.mask_degraded_ellipsoids(T(sc[:, 0]), T(sc[:, 1]), T(sc[:, 2]))
    out = {"a2_scale": sc, "a2_mask": N(mask)}
    # a6/a7 at the three target counts; scales as the synthetic scene + a few elongated ones
    E = 64
    scale = (0.005 + 0.05 * rng.random((E, 3))).astype(np.float32)
    scale[:6] = np.exp(rng.normal(-3.0, 1.0, size=(6, 3))).astype(np.float32)
    keep = N(q.mask_degraded_ellipsoids(T(scale[:, 0]), T(scale[:, 1]), T(scale[:, 2])))
    scale = scale[keep]
    E = scale.shape[0]
    rot = rng.standard_normal((E, 4)).astype(np.float32)
    xyz = rng.standard_normal((E, 3)).astype(np.float32)
    nrm = rng.standard_normal((E, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    Rm = gu.build_rotation(torch.nn.functional.normalize(T(rot)))
    L = gu.build_scaling_rotation(T(scale), T(rot))
    cov = L @ L.transpose(1, 2)
    out.update(scale=scale, rot=rot, xyz=xyz, normals=nrm, rotmat=N(Rm), cov=N(cov))
    for P in (50, 64, 256):
        pts, eid = q.compute_quadricell_centers(T(scale[:, 0]), T(scale[:, 1]), T(scale[:, 2]), target_points=P)
        ori, dr, mid = q.mask_and_compute_rays(pts, eid, T(nrm), T(xyz), cov, Rm, direction_mode="isocell")
        out[f"P{P}_points"] = N(pts)
        out[f"P{P}_eid"] = N(eid)
        out[f"P{P}_ori"] = N(ori)
        out[f"P{P}_dir"] = N(dr)
        out[f"P{P}_mid"] = N(mid)
        print(f"  P={P}: cells={pts.shape[0]} rays={ori.shape[0]}")
    save("g1_quadricell", **out)


# --------------------------------------------------------------------------------------
# g2: a4 compute_normals, a5 sym_eig_3x3
# --------------------------------------------------------------------------------------
def g2(m):
    s, se = m["sampling"], m["sym_eig_3x3"]
    rng = np.random.default_rng(22)
    pts = rng.standard_normal((300, 3)).astype(np.float32)
    nrm = s.compute_normals(T(pts), T(pts), k_neighbors=20)
    # neighbour sets as the reference's cdist+topk saw them (for the tie policy)
    d = torch.cdist(T(pts)[None], T(pts)[None], p=2.0)
    _, knn = torch.topk(d, 20, dim=2, largest=False)
    a = rng.standard_normal((200, 3, 3)).astype(np.float32)
    spd = a @ a.transpose(0, 2, 1)
    diag = np.zeros((8, 3, 3), np.float32)
    for i in range(8):
        diag[i] = np.diag(rng.random(3).astype(np.float32) * (i + 1))
    rep = np.zeros((4, 3, 3), np.float32)
    rep[0] = np.eye(3) * 2.0
    rep[1] = np.diag([1.0, 1.0, 3.0])
    qm = np.linalg.qr(rng.standard_normal((3, 3)))[0].astype(np.float32)
    rep[2] = (qm @ np.diag([2.0, 2.0, 5.0]).astype(np.float32) @ qm.T)
    rep[3] = (qm @ np.diag([1.0, 4.0, 4.0]).astype(np.float32) @ qm.T)
    rep = 0.5 * (rep + rep.transpose(0, 2, 1))
    mats = np.concatenate([spd, diag, rep], 0).astype(np.float32)
    vals, vecs = se.sym_eig_3x3(T(mats), eigenvectors=True)
    save("g2_normals", pts=pts, normals=N(nrm), knn=N(knn[0]), mats=mats, eigvals=N(vals), eigvecs=N(vecs))


# --------------------------------------------------------------------------------------
# g3: a8 isocell_distribution, a9 rotate_isocell
# --------------------------------------------------------------------------------------
def g3(m):
    iso = m["isocell"]
    rng = np.random.default_rng(33)
    out = {}
    for tgt, n0 in ((35, 3), (64, 1), (256, 1), (50, 1)):
        d = iso.isocell_distribution(tgt, torch.float32, "cpu", N0=n0, isrand=-1)
        out[f"dirs_{tgt}_{n0}"] = N(d)
    nrm = rng.standard_normal((12, 3)).astype(np.float32)
    nrm[0] = [0, 0, 1.0]          # parallel to z: s = 0 -> NaN in the reference (isocell.py:208-212)
    nrm[1] = [1e-4, 0, 1.0]       # near z
    nrm[2] = [0, 0, -1.0]         # anti-parallel
    nrm[3] = [3.0, 0, 0]          # un-normalised input
    out["normals"] = nrm
    for tgt in (64, 256):
        rd = iso.rotate_isocell(T(out[f"dirs_{tgt}_1"]), T(nrm))
        out[f"rot_{tgt}"] = N(rd)
    save("g3_isocell", **out)


# --------------------------------------------------------------------------------------
# g4: a10 eval_sh / evaluate_viewdirs_color
# --------------------------------------------------------------------------------------
def g4(m):
    s = m["sampling"]
    rng = np.random.default_rng(44)
    R = 512
    sh = (1.5 * rng.standard_normal((R, 3, 16))).astype(np.float32)  # large enough to hit clamp_min
    d = rng.standard_normal((R, 3))
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    out = {"sh": sh, "dir": d}
    for deg in range(4):
        out[f"rgb_deg{deg}"] = N(s.evaluate_viewdirs_color(T(sh), T(d), deg))
    save("g4_sh", **out)


# --------------------------------------------------------------------------------------
# g5: a12-a15 ray MLP, attention, scores, topk
# --------------------------------------------------------------------------------------
def _load_scorer(m, sd_np):
    idm = m["identification_module"].IdentificationModule("dino")
    sd = {k: T(v) for k, v in sd_np.items()}
    missing, unexpected = idm.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("backbone_wrapper.image_preprocessing_net") or k.startswith("camera_direction")
               or k.startswith("backbone_wrapper.norm") for k in missing), missing
    idm.eval()
    return idm


def g5(m):
    sd_np = syn.make_scorer_state_dict(0)
    idm = _load_scorer(m, sd_np)
    out = {"sd_checksum": np.int64(syn.checksum(sd_np))}
    R = 4096
    rays = syn.make_rays(R, 0)
    ori, dr, rgb = T(rays["ori"]), T(rays["dir"]), T(rays["rgb"])
    with torch.no_grad():
        pe = m["ray_preprocessor"].positional_encoding(ori[:16], 8)
        feat = idm.ray_preprocessor(ori, dr, rgb)
        kk = idm.attention.k_proj(feat)
        out.update(pe_pts16=N(pe), feat_head=N(feat[:128]), key_head=N(kk[:128]),
                   feat_sum=N(feat.double().sum(0)), key_sum=N(kk.double().sum(0)))
        # (T, token scale): flat regime = raw random init, peaky regime = scaled tokens
        for tag, tcount, tscale in (("flat256", 256, 1.0), ("peaky256", 256, 40.0), ("peaky137", 137, 40.0),
                                    ("mid1", 1, 10.0)):
            tok = syn.make_tokens(tcount, 1, tscale)
            att = idm.attention(T(tok), feat)
            score = att.sum(0)
            top = torch.topk(score, 100)
            q = idm.attention.q_proj(T(tok))
            # fp64 scores for the near-tie policy
            logits = (q.double() @ kk.double().T) / np.sqrt(384.0)
            score64 = torch.softmax(logits, -1).sum(0)
            out[f"{tag}_q_head"] = N(q[:8])
            out[f"{tag}_scores"] = N(score)
            out[f"{tag}_scores64"] = N(score64)
            out[f"{tag}_idx"] = N(top.indices)
            out[f"{tag}_val"] = N(top.values)
            out[f"{tag}_rowmax"] = N((logits.float()).max(-1).values)
            s64 = torch.sort(score64, descending=True).values
            gap = ((s64[:-1] - s64[1:]) / s64[:-1])[:101]
            print(f"  {tag}: top1={s64[0]:.4e} med={score64.median():.4e} min rel gap(top101)={gap.min():.2e} "
                  f"gap100/101={gap[99]:.2e} set==fp64: {set(N(top.indices).tolist()) == set(N(torch.topk(score64, 100).indices).tolist())}")
    save("g5_scorer", **out)


# --------------------------------------------------------------------------------------
# g6: a17-a21 filter, line intersection, exclude_negatives, rotation, errors
# --------------------------------------------------------------------------------------
def _pose_tail(m, ori, dr, idx, weights, up, gt_c2w):
    """The tail of the per-image loop, reference test.py:157-198,216-218,268-288, executed through
    the reference's own functions."""
    li, ec = m["line_intersection"], m["error_computation"]
    unique_elements, counts = torch.unique(ori[idx], return_counts=True, dim=0)
    mask = torch.isin(ori[idx], unique_elements[counts == 1], assume_unique=True).any(dim=1)
    idx = idx[mask]
    weights = weights[mask]
    weights = torch.divide(weights, torch.sum(weights))
    c0 = li.compute_line_intersection_impl2(ori[idx], dr[idx])
    weights = torch.multiply(weights, li.exclude_negatives(c0, ori[idx], dr[idx]))
    weights = torch.divide(weights, torch.sum(weights))
    c1 = li.compute_line_intersection_impl2(ori[idx], dr[idx])
    watch = torch.multiply(dr[idx], weights[:, None]).sum(dim=0)
    watch = torch.divide(watch, torch.linalg.norm(watch, dim=-1, keepdim=True))
    c2w = torch.eye(4, dtype=ori.dtype)
    w2c_R = li.make_rotation_mat(-watch, up)
    singular_rot = bool(torch.linalg.det(w2c_R) < 1.0e-7)
    if singular_rot:
        w2c_R = torch.eye(3)
    c2w[:3, :3] = torch.linalg.inv(w2c_R)
    c2w[:3, -1] = c1
    nan_pose = bool(torch.isnan(c2w).any())
    if nan_pose:
        c2w = torch.eye(4, dtype=ori.dtype)
    hom = torch.tensor([0.0, 0.0, 0.0, 1.0]).reshape(1, 4)
    terr = ec.compute_translation_error(hom @ gt_c2w[:3, :].T, hom @ c2w[:3, :].T)
    aerr = ec.compute_angular_error(gt_c2w[:3, :3], c2w[:3, :3])
    return dict(keep_mask=N(mask), idx_f=N(idx), w_final=N(weights), centre0=N(c0), centre=N(c1), watch=N(watch),
                c2w=N(c2w), flags=np.array([singular_rot, nan_pose]), terr=N(terr), aerr=N(aerr))


def g6(m):
    rng = np.random.default_rng(66)
    out = {}
    R = 2000

    def camera_case(name, n_dup=0, n_behind=0, parallel=False, up=None, weights_equal=False, k=100, few_unique=0):
        centre = rng.standard_normal(3).astype(np.float32) * 2.0
        ori = (rng.standard_normal((R, 3)) * 1.5).astype(np.float32)
        dr = rng.standard_normal((R, 3)).astype(np.float32)
        dr /= np.linalg.norm(dr, axis=-1, keepdims=True)
        idx = rng.permutation(R)[:k].astype(np.int64)
        # make the selected rays point at the centre (+noise)
        tgt = centre[None] - ori[idx] + 0.02 * rng.standard_normal((k, 3)).astype(np.float32)
        dsel = tgt / np.linalg.norm(tgt, axis=-1, keepdims=True)
        if parallel:
            # exactly axis-aligned: I - d d^T = diag(1,1,0), the normal matrix is exactly singular
            dsel[:] = np.array([0.0, 0.0, 1.0], np.float32)
        dr[idx] = dsel.astype(np.float32)
        for j in range(n_behind):
            dr[idx[3 + j]] = -dr[idx[3 + j]]
        for j in range(n_dup):       # duplicate origins among the selected rays (same ellipsoid surface point)
            ori[idx[10 + 2 * j + 1]] = ori[idx[10 + 2 * j]]
        if n_dup:                    # one origin sharing a single coordinate value with a unique one (isin quirk)
            ori[idx[40], 0] = ori[idx[41], 1]
            ori[idx[30]] = ori[idx[10]]   # a triple
        if few_unique:               # only `few_unique` origins occur once -> torch.isin takes its non-sorting path
            for j in range(few_unique, k):
                ori[idx[j]] = ori[idx[few_unique + (j - few_unique) // 2 * 2]]
        w = np.sort(rng.random(k).astype(np.float32) + 0.5)[::-1].copy()
        if weights_equal:
            w[:] = 1.0
        upv = np.array([0.1, 0.9, 0.2], np.float32) if up is None else np.asarray(up, np.float32)
        upv = upv / np.linalg.norm(upv)
        gt = np.eye(4, dtype=np.float32)
        gt[:3, :3] = syn.random_rotation(rng).astype(np.float32)
        gt[:3, 3] = centre
        res = _pose_tail(m, T(ori), T(dr), T(idx), T(w), T(upv), T(gt))
        out.update({f"{name}_ori": ori, f"{name}_dir": dr, f"{name}_idx": idx, f"{name}_w": w,
                    f"{name}_up": upv, f"{name}_gt": gt})
        out.update({f"{name}_{k}": v for k, v in res.items()})
        print(f"  {name}: kept={res['idx_f'].shape[0]} centre={res['centre']} flags={res['flags']} terr={res['terr']:.3e} aerr={res['aerr']:.3e}")

    camera_case("plain")
    camera_case("dups", n_dup=5)
    camera_case("behind", n_behind=7)
    camera_case("parallel", parallel=True)
    camera_case("mixed", n_dup=3, n_behind=4, weights_equal=True)
    out["cases"] = np.array(["plain", "dups", "behind", "parallel", "mixed"])
    # watch dir (anti)parallel to up -> singular rotation branch
    camera_case("upsing", up=None)
    camera_case("k37", k=37, n_behind=2)
    camera_case("few5", k=31, few_unique=5)
    camera_case("few8", k=32, few_unique=8)
    out["cases"] = np.array(["plain", "dups", "behind", "parallel", "mixed", "upsing", "k37", "few5", "few8"])
    # a20 make_rotation_mat alone (line_intersection.py:5-26), incl. up parallel to the direction (NaN)
    li = m["line_intersection"]
    dirs = rng.standard_normal((16, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    ups = rng.standard_normal((16, 3)).astype(np.float32)
    ups /= np.linalg.norm(ups, axis=-1, keepdims=True)
    ups[0] = dirs[0]
    ups[1] = -dirs[1]
    out["rot_dirs"], out["rot_ups"] = dirs, ups
    out["rot_mats"] = np.stack([N(li.make_rotation_mat(T(d), T(u))) for d, u in zip(dirs, ups)])
    save("g6_pose", **out)


# --------------------------------------------------------------------------------------
# g7: a11 generate_all_possible_rays (captured permutation) + a24 test_pose_estimation
# --------------------------------------------------------------------------------------
def _make_model(m, scene):
    gm = m["gaussian_model"].GaussianModel(int(scene["sh_degree"]))
    gm._xyz = T(scene["xyz"])
    gm._scaling = T(scene["log_scale"])
    gm._rotation = T(scene["rot"])
    gm._features_dc = T(scene["f_dc"])
    gm._features_rest = T(scene["f_rest"])
    gm._opacity = T(scene["opacity"])
    gm.active_sh_degree = int(scene["sh_degree"])
    return gm


def g7(m):
    s = m["sampling"]
    out = {}
    perms = []
    orig_randperm = torch.randperm

    def rec_randperm(*a, **k):
        p = orig_randperm(*a, **k)
        perms.append(p.clone())
        return p

    for tag, n, P, seed in (("n3000_p50", 3000, 50, 3), ("n400_p64", 400, 64, 4)):
        scene = syn.make_scene(n, seed)
        gm = _make_model(m, scene)
        perms.clear()
        torch.manual_seed(123)
        torch.randperm = rec_randperm
        try:
            ori, dr, rgb = s.generate_all_possible_rays(gm, sample_quadricell_targets=P)
        finally:
            torch.randperm = orig_randperm
        assert len(perms) == 1
        out[f"{tag}_perm"] = N(perms[0])
        out[f"{tag}_ori"], out[f"{tag}_dir"], out[f"{tag}_rgb"] = N(ori), N(dr), N(rgb)
        out[f"{tag}_meta"] = np.array([n, P, seed])
        print(f"  rays {tag}: R={ori.shape[0]}")

    # a24: the loop itself, 3 cameras, loss_fn=None (true inference; pretrain_eval_attention.py:136)
    sd_np = syn.make_scorer_state_dict(0, with_cnn=True)
    idm = _load_scorer(m, sd_np)
    CameraInfo = m["scene_structure"].CameraInfo
    cams_np = syn.make_cameras(3, 7, width=96, height=96, rgba=False) + syn.make_cameras(1, 8, width=80, height=80, rgba=True)
    cams = [CameraInfo(**c) for c in cams_np]
    rays_ori, rays_dir, rays_rgb = T(out["n3000_p50_ori"]), T(out["n3000_p50_dir"]), T(out["n3000_p50_rgb"])
    # capture the boundary inputs (tokens, fmap->up) per image through a hook on run_attention
    captured = []
    orig_run = idm.run_attention

    def run_att(img, mask, ro, rd, rr):
        f_pe, f_flat, fmap = idm.backbone_wrapper(img, mask)
        score, att, f_flat2, up = orig_run(img, mask, ro, rd, rr)
        captured.append(dict(tokens=N(f_pe), fmap=N(fmap), up=N(up), scores=N(score)))
        return score, att, f_flat2, up

    idm.run_attention = run_att
    # scale q so that the scores are peaky enough for a stable top-100 (documented in tests)
    with torch.no_grad():
        idm.attention.q_proj.weight.mul_(40.0)
    results, te, ae, _, _ = m["test"].test_pose_estimation(
        cams, idm, rays_ori, rays_dir, rays_rgb, torch.tensor([0.0, 1.0, 0.0]), loss_fn=None)
    out["e2e_q_scale"] = np.float32(40.0)
    out["e2e_sd_checksum"] = np.int64(syn.checksum(sd_np))
    for i, (r, c) in enumerate(zip(results, captured)):
        out[f"e2e{i}_tokens"] = c["tokens"]
        out[f"e2e{i}_up"] = c["up"]
        out[f"e2e{i}_fmap"] = c["fmap"].astype(np.float32)
        out[f"e2e{i}_scores"] = c["scores"]
        out[f"e2e{i}_pred_c2w"] = np.array(r["pred_c2w"], np.float32)
        out[f"e2e{i}_gt_c2w"] = np.array(r["gt_c2w"], np.float32)
        out[f"e2e{i}_loss"] = np.float32(r["loss"])
    out["e2e_mean_terr"] = np.float64(te)
    out["e2e_mean_aerr"] = np.float64(ae)
    out["e2e_n"] = np.int64(len(results))
    save("g7_e2e", **out)


# --------------------------------------------------------------------------------------
# g8: a24 with loss_fn (test.py:108-142): the reference loop with the stand-in callable of oracle/standin_loss.py
# --------------------------------------------------------------------------------------
def g8(m):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from standin_loss import line_distance_loss
    g7d = np.load(os.path.join(OUT, "g7_e2e.npz"))
    sd_np = syn.make_scorer_state_dict(0, with_cnn=True)
    idm = _load_scorer(m, sd_np)
    CameraInfo = m["scene_structure"].CameraInfo
    cams_np = syn.make_cameras(3, 7, width=96, height=96, rgba=False) + syn.make_cameras(1, 8, width=80, height=80, rgba=True)
    cams = [CameraInfo(**c) for c in cams_np]
    rays_ori, rays_dir, rays_rgb = T(g7d["n3000_p50_ori"]), T(g7d["n3000_p50_dir"]), T(g7d["n3000_p50_rgb"])
    with torch.no_grad():
        idm.attention.q_proj.weight.mul_(40.0)
    results, te, ae, ls, rc = m["test"].test_pose_estimation(
        cams, idm, rays_ori, rays_dir, rays_rgb, torch.tensor([0.0, 1.0, 0.0]), loss_fn=line_distance_loss)
    out = {"n": np.int64(len(results)), "mean_terr": np.float64(te), "mean_aerr": np.float64(ae), "mean_loss": np.float64(ls),
           "mean_recall": np.float64(rc)}
    for i, r in enumerate(results):
        out[f"r{i}_pred_c2w"] = np.array(r["pred_c2w"], np.float32)
        out[f"r{i}_loss"] = np.float32(r["loss"])
        out[f"r{i}_scores_loss"] = np.float64(r["scores_loss"])
        out[f"r{i}_recall"] = np.float64(r["recall"])
    print("  g8: terr %.3e aerr %.3e loss %.3e recall %.3f" % (te, ae, ls, rc))
    save("g8_lossfn", **out)


# --------------------------------------------------------------------------------------
# g9: DistanceBasedScoreLoss.forward (distance_based_loss.py:147-283) on the g7 rays
# --------------------------------------------------------------------------------------
def g9(m):
    dl = importlib.import_module("pose_estimation.distance_based_loss")
    g7d = np.load(os.path.join(OUT, "g7_e2e.npz"))
    ori, dr = T(g7d["n3000_p50_ori"]), T(g7d["n3000_p50_dir"])
    rng = np.random.default_rng(21)
    out = {"n": np.int64(4)}
    loss = dl.DistanceBasedScoreLoss()
    inside = np.eye(4, dtype=np.float32)                     # a camera INSIDE the scene: half of the origins lie behind its plane
    inside[:3, :3] = syn.random_rotation(np.random.default_rng(5))
    inside[:3, 3] = [0.1, -0.2, 0.05]
    for i in range(4):
        pose = T(g7d[f"e2e{i}_gt_c2w"]) if i < 3 else T(inside)
        K = T(np.array([[110.0, 0, 48], [0, 110.0, 48], [0, 0, 1]], np.float32))
        pred = T((rng.random(ori.shape[0]) * 0.02).astype(np.float32))
        ntok = [256, 137, 1, 200][i]
        avg, comb = loss(pred, pose, K, ori, dr, ntok, (16, 16), model_up=None, obs_img_shape=(96, 96))
        out[f"c{i}_pose"], out[f"c{i}_pred"], out[f"c{i}_ntok"] = N(pose), N(pred), np.int64(ntok)
        out[f"c{i}_combined"], out[f"c{i}_loss"] = N(comb), np.float64(avg.item())
        print(f"  g9 case {i}: loss {avg.item():.6e}  nonzero targets {int((comb > 0).sum())}/{comb.shape[0]}")
    save("g9_distance_loss", **out)


# --------------------------------------------------------------------------------------
# g10: one accumulated TRAINING step of the reference (train.py:106-170): forward, DistanceBasedScoreLoss + 0.1 x camera-up
# loss, backward -- gradient norms and 512 sampled entries per parameter tensor
# --------------------------------------------------------------------------------------
def g10(m):
    dl = importlib.import_module("pose_estimation.distance_based_loss")
    g7d = np.load(os.path.join(OUT, "g7_e2e.npz"))
    sd_np = syn.make_scorer_state_dict(0, with_cnn=True)
    idm = _load_scorer(m, sd_np)
    idm.train()
    idm.backbone_wrapper.eval()
    ori, dr, rgb = T(g7d["n3000_p50_ori"][:6000]), T(g7d["n3000_p50_dir"][:6000]), T(g7d["n3000_p50_rgb"][:6000])
    tok_pe, fmap = T(g7d["e2e0_tokens"]), T(g7d["e2e0_fmap"])
    tok_flat = fmap.permute(1, 2, 0).reshape(-1, fmap.shape[0])
    idm.backbone_wrapper.forward = lambda img, mask: (tok_pe, tok_flat, fmap)     # the boundary's image-side inputs, injected
    perms = []
    orig = torch.randperm

    def rec(*a, **k):
        p_ = orig(*a, **k)
        perms.append(p_.clone())
        return p_

    torch.manual_seed(77)
    torch.randperm = rec
    try:
        scores, att, _, up, used = idm(torch.zeros(8, 8, 3), torch.ones(8, 8, dtype=torch.bool), ori, dr, rgb)
    finally:
        torch.randperm = orig
    pose = T(g7d["e2e0_gt_c2w"])
    K = T(np.array([[110.0, 0, 48], [0, 110.0, 48], [0, 0, 1]], np.float32))
    model_up = T(np.array([0.0, 1.0, 0.0], np.float32))
    loss_score, _ = dl.DistanceBasedScoreLoss()(scores, pose, K, ori[used], dr[used], att.shape[-2], idm.backbone_wrapper.backbone_wh, model_up=model_up)
    cam_up = -0.5 * torch.cosine_similarity(model_up, up, dim=-1) + 0.5
    combined = loss_score + 0.1 * cam_up
    (combined / 32).backward()
    out = {"perm": N(perms[0]), "scores": N(scores), "loss_score": np.float64(loss_score.item()), "cam_up": np.float64(cam_up.item()),
           "combined": np.float64(combined.item()), "up": N(up), "n_rays": np.int64(6000)}
    names = []
    for name, p_ in idm.named_parameters():
        if p_.grad is None or name.startswith("backbone_wrapper"):
            continue
        g = p_.grad.detach().reshape(-1).double().numpy()
        idx = np.random.default_rng(abs(hash(name)) % (2 ** 31) if False else len(name) * 1009 + g.size).integers(0, g.size, 512)
        out["gn_" + name] = np.float64(np.linalg.norm(g))
        out["gi_" + name] = idx.astype(np.int64)
        out["gv_" + name] = g[idx].astype(np.float32)
        names.append(name)
    out["names"] = np.array(names)
    print(f"  g10: combined {combined.item():.6e}  {len(names)} parameter gradients")
    save("g10_train_step", **out)


# --------------------------------------------------------------------------------------
# g11: a23 BackboneWrapper.forward with structured alpha masks (mask -> token selection, backbone.py:86-114)
# --------------------------------------------------------------------------------------
def g11(m):
    idm = _load_scorer(m, syn.make_scorer_state_dict(0, with_cnn=True))
    out = {}
    for i, cam in enumerate(syn.make_masked_cameras(9, 120)):
        # test.py:69-83 literally
        obs = torch.from_numpy(np.array(cam["image"])).to(dtype=torch.float32) / 255.0
        mask = obs[..., -1] > 0.3
        obs = torch.multiply(obs[..., :3], obs[..., -1:]) + (1 - obs[..., -1:])
        with torch.no_grad():
            f_pe, f_flat, fmap = idm.backbone_wrapper(obs, mask)
        out[f"m{i}_tokens"], out[f"m{i}_flat"], out[f"m{i}_fmap"] = N(f_pe), N(f_flat), N(fmap).astype(np.float32)
        print(f"  masked camera {i}: {f_pe.shape[0]} of 256 tokens kept")
    save("g11_backbone_masks", **out)


# --------------------------------------------------------------------------------------
# g12: dataset camera loaders (scene/dataset_loader.py:5-20 -> colmap.py, tanksandtemples.py, synthetic.py) on the tiny on-disk
# scenes of synthetic.write_dataset_fixtures, and experiment discovery (pose_estimation/file_utils.py:19-72)
# --------------------------------------------------------------------------------------
def g12(m):
    import tempfile
    dl = importlib.import_module("scene.dataset_loader")
    # file_utils imports the ANTLR-generated cfg_grammar (antlr4 is absent here): stub the import, only the discovery functions are used
    cg = types.ModuleType("cfg_grammar")
    cg.parse_config = lambda text: (_ for _ in ()).throw(RuntimeError("antlr4 absent"))
    sys.modules.setdefault("cfg_grammar", cg)
    fu = importlib.import_module("pose_estimation.file_utils")
    out = {}
    # Environment shim: the T&T / Blender readers call Image.fromarray(int8 array, "RGB") (tanksandtemples.py:64, synthetic.py:49).  The
    # Pillow the reference pins (environment.yml) takes the buffer as raw bytes when a mode is given; the Pillow of this image refuses
    # int8.  Restore the pinned behaviour: same bytes, viewed as uint8.
    from PIL import Image as PILImage
    orig_fromarray = PILImage.fromarray

    def fromarray_compat(obj, mode=None):
        a = np.asarray(obj)
        return orig_fromarray(a.view(np.uint8) if (mode is not None and a.dtype == np.int8) else a, mode)

    PILImage.fromarray = fromarray_compat
    with tempfile.TemporaryDirectory() as td:
        srcs = syn.write_dataset_fixtures(td, 0)
        for name, src in srcs.items():
            for ev, wb in ((True, False), (False, True)):
                args = fu.dotdict(source_path=src, images=None, eval=ev, white_background=wb)
                info = dl.load_data(args)
                tag = f"{name}_e{int(ev)}w{int(wb)}"
                for split, cams in (("train", info.train_cameras), ("test", info.test_cameras)):
                    out[f"{tag}_{split}_n"] = np.int64(len(cams))
                    if not cams:
                        continue
                    out[f"{tag}_{split}_RT"] = np.stack([np.concatenate([np.asarray(c.R, np.float64).reshape(9), np.asarray(c.T, np.float64),
                                                                         [c.FovY, c.FovX, c.width, c.height, c.uid]]) for c in cams])
                    out[f"{tag}_{split}_names"] = np.array([c.image_name for c in cams])
                    out[f"{tag}_{split}_img"] = np.stack([np.array(c.image) for c in cams])
                out[f"{tag}_radius"] = np.float64(info.nerf_normalization["radius"])
                out[f"{tag}_translate"] = np.asarray(info.nerf_normalization["translate"], np.float64)
                out[f"{tag}_prefix"] = np.array(dl.get_dataset_prefix(src))
                print(f"  {tag}: {len(info.train_cameras)} train / {len(info.test_cameras)} test cameras")
        # experiment discovery
        exp = os.path.join(td, "output")
        for d, its in (("mip_360_garden_ab12", (7000, 30000)), ("mip_360_room_cd34", (30000,)), ("tt_Ignatius_ef56", (100, 20)), ("tt_empty_gh78", ()),
                       ("synthetic_lego_0001", (5,))):
            os.makedirs(os.path.join(exp, d, "point_cloud"), exist_ok=True)
            for it in its:
                os.makedirs(os.path.join(exp, d, "point_cloud", f"iteration_{it}"), exist_ok=True)
                open(os.path.join(exp, d, "point_cloud", f"iteration_{it}", "point_cloud.ply"), "wb").close()
        os.makedirs(os.path.join(exp, "tt_Ignatius_ef56", "point_cloud", "iteration_900"), exist_ok=True)      # no ply inside: not valid
        os.makedirs(os.path.join(exp, "mip_360_room_cd34", "point_cloud", "notes_1"), exist_ok=True)
        for prefix in ("", "mip_360_", "tt_", "synthetic_"):
            found = fu.parse_exp_dir(exp, prefix)
            out[f"exp_{prefix or 'all'}"] = np.array([f"{k}|{v['category_name']}|{os.path.relpath(v['checkpoint_filepath'], exp)}" for k, v in found.items()])
    save("g12_datasets", **out)


# --------------------------------------------------------------------------------------
# g13: what is left of pose_estimation/isocell.py -- the random modes of isocell_distribution (isrand 1..4, :47-61) and the grouping
# helpers group_by_360_isocell / get_dirs_group_idx (:87-154).  Neither is called anywhere in the reference.  The random modes add a th0 of
# shape [n rings] to a vector of shape [N0 n^2 cells] (:24,44): they raise RuntimeError for every target with more than one ring and only run
# for n = 1 -- recorded here: which (target, N0, mode) raise, and the n = 1 outputs under a CPU seed.
# --------------------------------------------------------------------------------------
def g13(m):
    iso = m["isocell"]
    out, raises = {}, []
    for mode in (1, 2, 3, 4, 0, 7):          # 0, 7: any other value of isrand takes the else-branch (:61-63): random start angle, centred cells
        for tgt, n0 in ((1, 1), (3, 3), (4, 1), (64, 1), (35, 3)):
            torch.manual_seed(1300 + 10 * mode + n0)
            try:
                d = iso.isocell_distribution(tgt, torch.float32, "cpu", N0=n0, isrand=mode)
                out[f"rand_m{mode}_t{tgt}_n{n0}"] = N(d)
                raises.append([mode, tgt, n0, 0])
            except RuntimeError:
                raises.append([mode, tgt, n0, 1])
    out["raises"] = np.asarray(raises, np.int64)
    # grouping helpers on the reference's own directions (both hemispheres) and on random unit vectors
    rng = np.random.default_rng(1313)
    for tgt, n0 in ((27, 3), (48, 3), (64, 1)):
        up = N(iso.isocell_distribution(tgt, torch.float32, "cpu", N0=n0, isrand=-1))
        dirs = np.concatenate([up, up * np.array([1, 1, -1], np.float32)], 0)
        rnd = rng.standard_normal((40, 3)).astype(np.float32)
        rnd /= np.linalg.norm(rnd, axis=1, keepdims=True)
        dirs = np.concatenate([dirs, rnd.astype(np.float32)], 0)
        grp, ring, cell = iso.group_by_360_isocell(T(dirs), tgt, N0=n0)
        out[f"grp_{tgt}_{n0}_dirs"] = dirs
        out[f"grp_{tgt}_{n0}_group"], out[f"grp_{tgt}_{n0}_ring"], out[f"grp_{tgt}_{n0}_cell"] = N(grp), N(ring), N(cell)
        groups = iso.get_dirs_group_idx(T(dirs), tgt, N0=n0)
        out[f"grp_{tgt}_{n0}_sizes"] = np.asarray([g.shape[0] for g in groups], np.int64)
        out[f"grp_{tgt}_{n0}_members"] = np.concatenate([N(g) for g in groups]) if groups else np.zeros(0, np.int64)
    save("g13_isocell_rest", **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    torch.set_num_threads(8)
    m = import_reference()
    gens = {"g1": g1, "g2": g2, "g3": g3, "g4": g4, "g5": g5, "g6": g6, "g7": g7, "g8": g8, "g9": g9, "g10": g10, "g11": g11, "g12": g12, "g13": g13}
    only = [x for x in args.only.split(",") if x]
    for k, fn in gens.items():
        if only and k not in only:
            continue
        print(f"[{k}]")
        fn(m)


if __name__ == "__main__":
    main()
