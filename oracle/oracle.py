"""ctypes/numpy front end of the CPU oracle (oracle/sixdgs_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- never by the product package ``6dgs_amd``.
Parity status of the oracle itself: pinned against tests/golden/g1..g7 (generated from the
reference by oracle/gen_golden.py); see tests/test_oracle_golden.py.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsixdgs_oracle.so")
_lib = None

c_f = C.POINTER(C.c_float)
c_i64 = C.POINTER(C.c_int64)
c_u8 = C.POINTER(C.c_uint8)
c_i32 = C.POINTER(C.c_int)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "sixdgs_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.o_quadricell_centers.restype = C.c_int64
        _lib.o_mask_and_compute_rays.restype = C.c_int64
        _lib.o_isocell_distribution.restype = C.c_int64
        _lib.o_num_threads.restype = C.c_int
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=c_f):
    return a.ctypes.data_as(t)


def num_threads() -> int:
    return lib().o_num_threads()


def set_num_threads(n: int):
    lib().o_set_num_threads(int(n))


def build_rotation(rot4):
    rot4 = _f(rot4)
    out = np.empty((rot4.shape[0], 3, 3), np.float32)
    lib().o_build_rotation(_p(rot4), C.c_int64(rot4.shape[0]), _p(out))
    return out


def total_rings(scale, target_points=50):
    scale = _f(scale)
    out = np.empty(scale.shape[0], np.int64)
    lib().o_total_rings(_p(scale), C.c_int64(scale.shape[0]), int(target_points), _p(out, c_i64))
    return out


def mask_degraded(scale, target_points=50):
    scale = _f(scale)
    out = np.empty(scale.shape[0], np.uint8)
    lib().o_mask_degraded(_p(scale), C.c_int64(scale.shape[0]), int(target_points), _p(out, c_u8))
    return out.astype(bool)


def quadricell_centers(scale, target_points=50, res=1000):
    scale = _f(scale)
    E = scale.shape[0]
    n = lib().o_quadricell_centers(_p(scale), C.c_int64(E), int(target_points), int(res), None, None)
    pts = np.empty((n, 3), np.float32)
    eid = np.empty(n, np.int64)
    lib().o_quadricell_centers(_p(scale), C.c_int64(E), int(target_points), int(res), _p(pts), _p(eid, c_i64))
    return pts, eid


def mask_and_compute_rays(points, eid, normals, centers, rotmat):
    points, normals, centers, rotmat = _f(points), _f(normals), _f(centers), _f(rotmat)
    eid = np.ascontiguousarray(eid, np.int64)
    n = points.shape[0]
    ori = np.empty((n, 3), np.float32)
    dr = np.empty((n, 3), np.float32)
    mid = np.empty(n, np.int64)
    r = lib().o_mask_and_compute_rays(_p(points), _p(eid, c_i64), C.c_int64(n), _p(normals), _p(centers), _p(rotmat),
                                      _p(ori), _p(dr), _p(mid, c_i64))
    return ori[:r].copy(), dr[:r].copy(), mid[:r].copy()


def sym_eig_3x3(mats, eigenvectors=True):
    mats = _f(mats).reshape(-1, 3, 3)
    vals = np.empty((mats.shape[0], 3), np.float32)
    vecs = np.empty((mats.shape[0], 3, 3), np.float32) if eigenvectors else None
    lib().o_sym_eig_3x3(_p(mats), C.c_int64(mats.shape[0]), _p(vals), _p(vecs) if eigenvectors else None)
    return vals, vecs


def compute_normals(chunk, cloud, k=20, return_knn=False):
    chunk, cloud = _f(chunk), _f(cloud)
    out = np.empty((chunk.shape[0], 3), np.float32)
    knn = np.empty((chunk.shape[0], k), np.int64)
    lib().o_compute_normals(_p(chunk), C.c_int64(chunk.shape[0]), _p(cloud), C.c_int64(cloud.shape[0]), int(k), _p(out),
                            _p(knn, c_i64))
    return (out, knn) if return_knn else out


def isocell_distribution(ray_target, n0=1):
    n = lib().o_isocell_distribution(int(ray_target), int(n0), None)
    out = np.empty((n, 3), np.float32)
    lib().o_isocell_distribution(int(ray_target), int(n0), _p(out))
    return out


def rotate_isocell(dirs, normals):
    dirs, normals = _f(dirs), _f(normals)
    out = np.empty((normals.shape[0], dirs.shape[0], 3), np.float32)
    lib().o_rotate_isocell(_p(dirs), C.c_int64(dirs.shape[0]), _p(normals), C.c_int64(normals.shape[0]), _p(out))
    return out


def eval_sh_color(sh, dirs, deg):
    sh, dirs = _f(sh), _f(dirs)
    assert sh.ndim == 3 and sh.shape[1] == 3
    out = np.empty((dirs.shape[0], 3), np.float32)
    lib().o_eval_sh_color(_p(sh), int(sh.shape[2]), _p(dirs), C.c_int64(dirs.shape[0]), int(deg), _p(out))
    return out


def ray_input(ori, dr, rgb):
    ori, dr, rgb = _f(ori), _f(dr), _f(rgb)
    out = np.empty((ori.shape[0], 141), np.float32)
    lib().o_ray_input(_p(ori), _p(dr), _p(rgb), C.c_int64(ori.shape[0]), _p(out))
    return out


def linear(x, w, b, relu=False):
    x, w, b = _f(x), _f(w), _f(b)
    y = np.empty((x.shape[0], w.shape[0]), np.float32)
    lib().o_linear(_p(x), C.c_int64(x.shape[0]), int(x.shape[1]), _p(w), _p(b), int(w.shape[0]), int(relu), _p(y))
    return y


def ray_features(ori, dr, rgb, sd, want_feat=True, want_key=True):
    """sd: state dict (numpy) with the reference key names."""
    ori, dr, rgb = _f(ori), _f(dr), _f(rgb)
    R = ori.shape[0]
    g = lambda k: _f(sd[k])
    ws = [g("ray_preprocessor.mlp.0.weight"), g("ray_preprocessor.mlp.0.bias"), g("ray_preprocessor.mlp.2.weight"),
          g("ray_preprocessor.mlp.2.bias"), g("ray_preprocessor.mlp2.0.weight"), g("ray_preprocessor.mlp2.0.bias"),
          g("ray_preprocessor.mlp2.2.weight"), g("ray_preprocessor.mlp2.2.bias"), g("attention.k_proj.weight"),
          g("attention.k_proj.bias")]
    feat = np.empty((R, 384), np.float32) if want_feat else None
    key = np.empty((R, 384), np.float32) if want_key else None
    lib().o_ray_features(_p(ori), _p(dr), _p(rgb), C.c_int64(R), *[_p(w) for w in ws],
                         _p(feat) if want_feat else None, _p(key) if want_key else None)
    return feat, key


def q_proj(tokens, sd):
    return linear(tokens, sd["attention.q_proj.weight"], sd["attention.q_proj.bias"])


def attention_scores(q, key, return_stats=False):
    q, key = _f(q), _f(key)
    T, R = q.shape[0], key.shape[0]
    scores = np.empty(R, np.float32)
    mx = np.empty(max(T, 1), np.float32)
    sm = np.empty(max(T, 1), np.float32)
    lib().o_attention_scores(_p(q), int(T), _p(key), C.c_int64(R), _p(scores), _p(mx), _p(sm))
    return (scores, mx[:T], sm[:T]) if return_stats else scores


def topk(scores, k=100):
    scores = _f(scores)
    k = min(int(k), scores.shape[0])
    idx = np.empty(k, np.int64)
    val = np.empty(k, np.float32)
    lib().o_topk(_p(scores), C.c_int64(scores.shape[0]), k, _p(idx, c_i64), _p(val))
    return idx, val


def unique_origin_filter(sel_ori):
    sel_ori = _f(sel_ori)
    keep = np.empty(sel_ori.shape[0], np.uint8)
    lib().o_unique_origin_filter(_p(sel_ori), int(sel_ori.shape[0]), _p(keep, c_u8))
    return keep.astype(bool)


def line_intersection(pts, dirs):
    pts, dirs = _f(pts), _f(dirs)
    out = np.empty(3, np.float32)
    lib().o_line_intersection(_p(pts), _p(dirs), int(pts.shape[0]), _p(out))
    return out


def make_rotation_mat(direction, up):
    direction, up = _f(direction), _f(up)
    out = np.empty((3, 3), np.float32)
    lib().o_make_rotation_mat(_p(direction), _p(up), _p(out))
    return out


def pose_from_topk(rays_ori, rays_dir, idx, weights, up):
    rays_ori, rays_dir, weights, up = _f(rays_ori), _f(rays_dir), _f(weights), _f(up)
    idx = np.ascontiguousarray(idx, np.int64)
    k = idx.shape[0]
    c2w = np.empty((4, 4), np.float32)
    centre = np.empty(3, np.float32)
    wf = np.empty(k, np.float32)
    keep = np.empty(k, np.uint8)
    flags = np.zeros(2, np.int32)
    nk = C.c_int(0)
    lib().o_pose_from_topk(_p(rays_ori), _p(rays_dir), _p(idx, c_i64), _p(weights), int(k), _p(up), _p(c2w), _p(centre),
                           _p(wf), _p(keep, c_u8), _p(flags, c_i32), C.byref(nk))
    return dict(c2w=c2w, centre=centre, w_final=wf, keep=keep.astype(bool), flags=flags.astype(bool), n_kept=nk.value)


def pose_errors(gt_c2w, pred_c2w):
    gt, pr = _f(gt_c2w), _f(pred_c2w)
    t = C.c_float(0)
    a = C.c_float(0)
    lib().o_pose_errors(_p(gt), _p(pr), C.byref(t), C.byref(a))
    return t.value, a.value


def distance_target(rays_ori, rays_dir, pose, n_tokens):
    """distance_based_loss.py target scores (sum = n_tokens) for the c2w `pose` [4,4]; returns (combined [R], raw sum)."""
    o, d, pz = _f(rays_ori), _f(rays_dir), _f(np.asarray(pose, np.float32).reshape(16))
    out = np.empty(o.shape[0], np.float32)
    s = C.c_float(0)
    lib().o_distance_target(_p(o), _p(d), C.c_int64(o.shape[0]), _p(pz), int(n_tokens), _p(out), C.byref(s))
    return out, s.value
