"""Tensor-level front end of the C ABI (include/sixdgs.h): PyTorch owns memory and streams, the HIP
library does the work.  Every function requires CUDA(ROCm) tensors -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import functools
import os
import threading
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import Profile, ScorerWeights, check

D = 384
MMA_DEFAULT, MMA_F32, MMA_BF16X6, MMA_F16X3, MMA_F16X3_L32 = -1, 0, 1, 2, 3
F16_MODES = (MMA_F16X3, MMA_F16X3_L32)     # scaled fp16 key planes; they differ in how the logits are stored between the passes
MMA_LIBRARY_DEFAULT = MMA_F16X3
_mma_mode = MMA_DEFAULT


def set_mma_mode(mode: int):
    """Select how the fp32 contractions run on the matrix cores (see SIXDGS_MMA_* in include/sixdgs.h)."""
    global _mma_mode
    assert mode in (MMA_DEFAULT, MMA_F32, MMA_BF16X6, MMA_F16X3, MMA_F16X3_L32)
    _mma_mode = mode


def get_mma_mode() -> int:
    return _mma_mode


def effective_mma_mode() -> int:
    """The mode MMA_DEFAULT resolves to inside the library."""
    return MMA_LIBRARY_DEFAULT if _mma_mode == MMA_DEFAULT else _mma_mode

RAY_IN_PAD = 144
TOK_IN = 398
MAX_TOKENS = 256


def _p(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


_tls = threading.local()     # .device: index of the device of the call being marshalled on this thread (set by _on_device / _need_gpu)


def _stream():
    """The current PyTorch stream OF THE TENSORS' DEVICE (not of whatever device is current in the process)."""
    dev = getattr(_tls, "device", None)
    return C.c_void_p(torch.cuda.current_stream(dev if dev is not None else torch.cuda.current_device()).cuda_stream)


def _first_device(objs):
    for a in objs:
        if torch.is_tensor(a):
            if a.is_cuda:
                return a.device
        elif isinstance(a, (PackedWeights, SelectStream)):
            if getattr(a, "device", None) is not None:      # None while the object is being constructed
                return a.device
        elif isinstance(a, (list, tuple)) and a and torch.is_tensor(a[0]) and a[0].is_cuda:
            return a[0].device
        elif isinstance(a, (str, torch.device)) and not isinstance(a, bool):
            try:
                d = torch.device(a)
            except (RuntimeError, TypeError, ValueError):
                continue
            if d.type == "cuda":
                return torch.device("cuda", d.index if d.index is not None else torch.cuda.current_device())
    return None


def _on_device(fn):
    """Every library call runs with the device of its tensor arguments current -- the kernels launch on the device that is current
    when the library is entered -- and puts the process's current device back afterwards (a call on cuda:1 tensors does not move
    the caller to cuda:1).  The device is kept per THREAD for _stream(), so concurrent callers on different devices do not see each
    other's choice."""
    @functools.wraps(fn)
    def scoped(*args, **kwargs):
        dev = _first_device(list(args) + list(kwargs.values()))
        prev = getattr(_tls, "device", None)
        try:
            if dev is None or not torch.cuda.is_available():
                return fn(*args, **kwargs)
            _tls.device = dev.index
            if dev.index == torch.cuda.current_device():
                return fn(*args, **kwargs)
            with torch.cuda.device(dev):
                return fn(*args, **kwargs)
        finally:
            _tls.device = prev
    return scoped


def _need_gpu(*ts):
    """Every tensor argument must be a GPU tensor, all on ONE device -- the device _on_device made current for the call."""
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("6dgs_amd: tensors must live on the GPU (no CPU fallback on the product path)")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"6dgs_amd: tensor arguments on different devices ({dev} and {t.device})")
    if dev is not None:
        _tls.device = dev.index


def _f32(t: torch.Tensor) -> torch.Tensor:
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _i64(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.int64).contiguous()


# ---------------------------------------------------------------------------------------------
# geometry
# ---------------------------------------------------------------------------------------------
@_on_device
def mask_degraded(log_scale: torch.Tensor, target_points: int = 50) -> torch.Tensor:
    log_scale = _f32(log_scale)
    _need_gpu(log_scale)
    n = log_scale.shape[0]
    out = torch.empty(n, dtype=torch.uint8, device=log_scale.device)
    check(_lib.load().sixdgs_mask_degraded(_p(log_scale), n, int(target_points), _p(out), _stream()), "mask_degraded")
    return out.bool()


@_on_device
def sym_eig_3x3(mats: torch.Tensor, eigenvectors: bool = True):
    mats = _f32(mats)
    _need_gpu(mats)
    shape = mats.shape[:-2]
    m = mats.reshape(-1, 3, 3)
    vals = torch.empty(m.shape[0], 3, device=m.device)
    vecs = torch.empty(m.shape[0], 3, 3, device=m.device) if eigenvectors else None
    check(_lib.load().sixdgs_sym_eig_3x3(_p(m), m.shape[0], _p(vals), _p(vecs), _stream()), "sym_eig_3x3")
    return vals.reshape(*shape, 3), (vecs.reshape(*shape, 3, 3) if eigenvectors else None)


KNN_GRID_FROM = 16384     # clouds at least this large take the grid search (identical results, O(E) instead of O(E^2))


@_on_device
def normals_knn(query: torch.Tensor, cloud: torch.Tensor, k: int = 20, return_knn: bool = False, method: str = "auto"):
    """a4.  method: "brute" (the reference's exhaustive search), "grid" (uniform-grid search, same neighbour lists and
    normals bit for bit) or "auto" (grid from KNN_GRID_FROM points)."""
    same = query is cloud
    query, cloud = _f32(query), _f32(cloud)
    if same:
        query = cloud                      # keeps the pointer identity the grid kernel uses to walk the queries in cell order
    _need_gpu(query, cloud)
    lib = _lib.load()
    nq, e = query.shape[0], cloud.shape[0]
    out = torch.empty(nq, 3, device=query.device)
    knn = torch.empty(nq, k, dtype=torch.int64, device=query.device) if return_knn else None
    if method not in ("auto", "brute", "grid"):
        raise ValueError(method)
    if method == "grid" or (method == "auto" and e >= KNN_GRID_FROM):
        ws = torch.empty(lib.sixdgs_normals_knn_grid_workspace_bytes(e), dtype=torch.uint8, device=query.device)
        check(lib.sixdgs_normals_knn_grid(_p(query), nq, _p(cloud), e, int(k), _p(out), _p(knn), _p(ws), ws.numel(), _stream()),
              "normals_knn_grid")
    else:
        check(lib.sixdgs_normals_knn(_p(query), nq, _p(cloud), e, int(k), _p(out), _p(knn), _stream()), "normals_knn")
    return (out, knn) if return_knn else out


@_on_device
def quadricell_centers(scale: torch.Tensor, target_points: int = 50, table_res: int = 1000):
    """a6 alone: (points[C,3], ellipsoid_id[C]) for activated semi axes scale[E,3]."""
    scale = _f32(scale)
    _need_gpu(scale)
    lib = _lib.load()
    e = scale.shape[0]
    dev = scale.device
    counts = torch.empty(max(e, 1), dtype=torch.int64, device=dev)
    offs = torch.empty(max(e, 1), dtype=torch.int64, device=dev)
    total = torch.zeros(2, dtype=torch.int64, device=dev)
    check(lib.sixdgs_quadricell_cell_counts(_p(scale), e, int(target_points), _p(counts), _p(offs), _p(total), _stream()),
          "quadricell_cell_counts")
    c = int(total[0].item())
    pts = torch.empty(c, 3, device=dev)
    eid = torch.empty(c, dtype=torch.int64, device=dev)
    if c:
        check(lib.sixdgs_quadricell_centers(_p(scale), e, int(target_points), int(table_res), _p(offs), _p(pts), _p(eid),
                                            _stream()), "quadricell_centers")
    return pts, eid


@_on_device
def emit_quadricell(xyz, scale, rot, f_dc, f_rest, sh_degree: int, sel: Optional[torch.Tensor], normals, target_points: int = 50,
                    table_res: int = 1000, scale_is_log: bool = True, want_rgb: bool = True):
    """a1+a6+a7+a10: returns ori[R,3], dir[R,3], rgb[R,3] (or None), src[R] (Gaussian id), n_cells."""
    xyz, scale, rot, normals = _f32(xyz), _f32(scale), _f32(rot), _f32(normals)
    _need_gpu(xyz, scale, rot, normals)
    lib = _lib.load()
    dev = xyz.device
    sel_t = _i64(sel) if sel is not None else None
    e = sel_t.shape[0] if sel_t is not None else xyz.shape[0]
    assert normals.shape[0] == e
    counts = torch.empty(max(e, 1), dtype=torch.int64, device=dev)
    offs = torch.empty(max(e, 1), dtype=torch.int64, device=dev)
    total = torch.zeros(2, dtype=torch.int64, device=dev)
    check(lib.sixdgs_emit_quadricell_count(_p(xyz), _p(scale), int(scale_is_log), _p(rot), _p(sel_t), e, _p(normals),
                                           int(target_points), int(table_res), _p(counts), _p(offs), _p(total), _stream()),
          "emit_quadricell_count")
    tot = total.tolist()  # the one host sync of the emitter (the reference syncs at sampling.py:145)
    r, n_cells = int(tot[0]), int(tot[1])
    ori = torch.empty(r, 3, device=dev)
    dr = torch.empty(r, 3, device=dev)
    src = torch.empty(r, dtype=torch.int64, device=dev)
    rgb = None
    n_coef = 1
    if want_rgb:
        f_dc, f_rest = _f32(f_dc), _f32(f_rest)
        n_coef = 1 + f_rest.shape[1]
        rgb = torch.empty(r, 3, device=dev)
    if r:
        check(lib.sixdgs_emit_quadricell_write(_p(xyz), _p(scale), int(scale_is_log), _p(rot), _p(f_dc if want_rgb else None),
                                               _p(f_rest if want_rgb else None), int(sh_degree), int(n_coef), _p(sel_t), e,
                                               _p(normals), int(target_points), int(table_res), _p(offs), _p(ori), _p(dr),
                                               _p(rgb), _p(src), _stream()), "emit_quadricell_write")
    return ori, dr, rgb, src, n_cells


@_on_device
def isocell_distribution(ray_target: int, n0: int = 1, device="cuda") -> torch.Tensor:
    lib = _lib.load()
    cnt = C.c_int64(0)
    check(lib.sixdgs_isocell_distribution(int(ray_target), int(n0), None, C.byref(cnt), _stream()), "isocell_distribution")
    out = torch.empty(cnt.value, 3, device=device)
    _need_gpu(out)
    check(lib.sixdgs_isocell_distribution(int(ray_target), int(n0), _p(out), C.byref(cnt), _stream()), "isocell_distribution")
    return out


@_on_device
def rotate_isocell(dirs: torch.Tensor, normals: torch.Tensor) -> torch.Tensor:
    dirs, normals = _f32(dirs), _f32(normals)
    _need_gpu(dirs, normals)
    out = torch.empty(normals.shape[0], dirs.shape[0], 3, device=dirs.device)
    check(_lib.load().sixdgs_rotate_isocell(_p(dirs), dirs.shape[0], _p(normals), normals.shape[0], _p(out), _stream()),
          "rotate_isocell")
    return out


@_on_device
def emit_isocell(xyz, scale, rot, f_dc, f_rest, sh_degree: int, sel, normals, dirs, scale_is_log: bool = True,
                 want_rgb: bool = True, want_src: bool = True):
    xyz, scale, rot, normals, dirs = _f32(xyz), _f32(scale), _f32(rot), _f32(normals), _f32(dirs)
    _need_gpu(xyz, scale, rot, normals, dirs)
    dev = xyz.device
    sel_t = _i64(sel) if sel is not None else None
    e = sel_t.shape[0] if sel_t is not None else xyz.shape[0]
    k = dirs.shape[0]
    ori = torch.empty(e * k, 3, device=dev)
    dr = torch.empty(e * k, 3, device=dev)
    rgb = torch.empty(e * k, 3, device=dev) if want_rgb else None
    src = torch.empty(e * k, dtype=torch.int64, device=dev) if want_src else None
    n_coef = 1
    if want_rgb:
        f_dc, f_rest = _f32(f_dc), _f32(f_rest)
        n_coef = 1 + f_rest.shape[1]
    check(_lib.load().sixdgs_emit_isocell(_p(xyz), _p(scale), int(scale_is_log), _p(rot), _p(f_dc if want_rgb else None),
                                          _p(f_rest if want_rgb else None), int(sh_degree), int(n_coef), _p(sel_t), e,
                                          _p(normals), _p(dirs), k, _p(ori), _p(dr), _p(rgb), _p(src), _stream()), "emit_isocell")
    return ori, dr, rgb, src


@_on_device
def eval_sh_color(sh: torch.Tensor, dirs: torch.Tensor, sh_degree: int) -> torch.Tensor:
    sh, dirs = _f32(sh), _f32(dirs)
    _need_gpu(sh, dirs)
    out = torch.empty(dirs.shape[0], 3, device=dirs.device)
    check(_lib.load().sixdgs_eval_sh_color(_p(sh), sh.shape[-1], _p(dirs), dirs.shape[0], int(sh_degree), _p(out), _stream()),
          "eval_sh_color")
    return out


# ---------------------------------------------------------------------------------------------
# scorer
# ---------------------------------------------------------------------------------------------
class PackedWeights:
    """Device copy of the scorer weights in the kernels' padded layout + the C struct of pointers."""

    KEYS = ("ray_preprocessor.mlp.0", "ray_preprocessor.mlp.2", "ray_preprocessor.mlp2.0", "ray_preprocessor.mlp2.2",
            "attention.k_proj", "attention.q_proj")

    @_on_device
    def __init__(self, state_dict, device):
        lib = _lib.load()
        srcs = []
        for k in self.KEYS:
            srcs.append(_f32(state_dict[k + ".weight"]).to(device))
            srcs.append(_f32(state_dict[k + ".bias"]).to(device))
        self._srcs = srcs
        self.buffer = torch.empty(lib.sixdgs_packed_weights_floats(), device=device)
        self.device = self.buffer.device
        self.struct = ScorerWeights()
        check(lib.sixdgs_pack_weights(*[_p(t) for t in srcs], _p(self.buffer), C.byref(self.struct), _stream()), "pack_weights")

    @property
    def ref(self):
        return C.byref(self.struct)


class KernelProfile:
    """Caller-owned HIP-event timing of the dominant kernel (see sixdgs_profile in include/sixdgs.h)."""

    def __init__(self):
        self.struct = Profile()
        self.struct.count = 0

    @property
    def ref(self):
        return C.byref(self.struct)

    def collect(self):
        """-> (milliseconds, algorithmic FLOP, algorithmic bytes, launches); waits for the recorded events."""
        ms, fl, by, n = C.c_double(0), C.c_double(0), C.c_double(0), C.c_int(0)
        check(_lib.load().sixdgs_profile_collect(self.ref, C.byref(ms), C.byref(fl), C.byref(by), C.byref(n)), "profile_collect")
        return ms.value, fl.value, by.value, n.value


@_on_device
def ray_encode(ori, dr, rgb) -> torch.Tensor:
    ori, dr, rgb = _f32(ori), _f32(dr), _f32(rgb)
    _need_gpu(ori, dr, rgb)
    x = torch.empty(ori.shape[0], RAY_IN_PAD, device=ori.device)
    check(_lib.load().sixdgs_ray_encode(_p(ori), _p(dr), _p(rgb), ori.shape[0], _p(x), _stream()), "ray_encode")
    return x


@_on_device
def linear(x, w, b=None, relu: bool = False, mma_mode: Optional[int] = None, split_k: Optional[int] = None) -> torch.Tensor:
    """y = x w^T + b (optionally ReLU).  split_k: None = automatic (K is cut into slices computed by separate workgroups when
    the output has too few 128x128 tiles to fill the 256 CUs and K is long), 1 = never, n = that many slices."""
    x, w = _f32(x), _f32(w)
    _need_gpu(x, w)
    b = _f32(b) if b is not None else None
    lib = _lib.load()
    m, k, n = x.shape[0], x.shape[1], w.shape[0]
    y = torch.empty(m, n, device=x.device)
    mode = _mma_mode if mma_mode is None else mma_mode
    if split_k is None:
        tiles = -(-m // 128) * -(-n // 128)
        split_k = 1 if (tiles >= 128 or k < 2048 or m == 0) else max(1, min(k // 256, -(-512 // tiles)))
    if split_k > 1:
        ws = torch.empty(lib.sixdgs_linear_splitk_workspace_bytes(m, n, split_k), dtype=torch.uint8, device=x.device)
        check(lib.sixdgs_linear_splitk(_p(x), m, k, x.stride(0), _p(w), w.stride(0), _p(b), n, int(relu), _p(y), y.stride(0), int(split_k),
                                       _p(ws), ws.numel(), _stream(), mode), "linear_splitk")
    else:
        check(lib.sixdgs_linear_ex(_p(x), m, k, x.stride(0), _p(w), w.stride(0), _p(b), n, int(relu), _p(y), y.stride(0), _stream(), mode),
              "linear")
    return y


TOK_ATTENTION_MAX_TOKENS = 288


@_on_device
def tok_attention(qkv: torch.Tensor, images: int, tokens: int, heads: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """softmax(q k^T / 8) v per image and head on the QKV product's output [images * tokens, 3 * heads * 64] as it lies (q | k | v, head h at columns
    h * 64 of each) -> [images * tokens, heads * 64], token-major (sixdgs_tok_attention; head dimension 64, tokens <= 288)."""
    qkv = qkv.detach()
    if qkv.dtype != torch.float32 or qkv.dim() != 2 or qkv.shape[0] != images * tokens or qkv.shape[1] != 3 * heads * 64 or qkv.stride(1) != 1:
        raise RuntimeError(f"6dgs_amd: tok_attention needs float32 qkv [{images * tokens}, {3 * heads * 64}], got {qkv.dtype} {tuple(qkv.shape)}")
    _need_gpu(qkv)
    y = out if out is not None else torch.empty(images * tokens, heads * 64, device=qkv.device)
    check(_lib.load().sixdgs_tok_attention(_p(qkv), qkv.stride(0), int(images), int(tokens), int(heads), _p(y), y.stride(0), _stream()), "tok_attention")
    return y


@_on_device
def im2col(x: torch.Tensor, k: int, taps_major: bool = False) -> torch.Tensor:
    """[B, C, H, W] fp32 (ANY strides: a permuted view of a [B*H*W, C] GEMM output is read in place) -> the A matrix [B*ho*wo, C*k*k] of the valid
    k x k convolution (sixdgs_im2col): columns in conv.weight.view(out, -1) order -- torch.nn.functional.unfold + transpose + contiguous in one launch --
    or, taps_major, in (ky, kx, c) order for a weight permuted as weight.permute(0, 2, 3, 1).reshape(out, -1) (plain 16-byte copies when the input's
    channel stride is 1)."""
    if x.dim() != 4 or x.dtype != torch.float32:
        raise RuntimeError(f"6dgs_amd: im2col needs a float32 [B, C, H, W] tensor, got {x.dtype} {tuple(x.shape)}")
    x = x.detach()
    _need_gpu(x)
    b, c, h, w = x.shape
    if h < k or w < k:
        raise RuntimeError(f"6dgs_amd: im2col: a {k} x {k} window does not fit a {h} x {w} map")
    a = torch.empty(b * (h - k + 1) * (w - k + 1), c * k * k, device=x.device)
    check(_lib.load().sixdgs_im2col(_p(x), x.stride(0), x.stride(1), x.stride(2), x.stride(3), b, c, h, w, int(k), 1 if taps_major else 0, _p(a), _stream()),
          "im2col")
    return a


@_on_device
def u8_to_planar(images_u8: torch.Tensor, table256: torch.Tensor) -> torch.Tensor:
    """[B, H, W, 3] uint8 -> [B, 3, H, W] fp32 = table256[value] (sixdgs_u8_to_planar; H*W a multiple of 4)."""
    if images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or images_u8.shape[-1] != 3 or (images_u8.shape[1] * images_u8.shape[2]) % 4:
        raise RuntimeError(f"6dgs_amd: u8_to_planar needs uint8 [B, H, W, 3] with H*W a multiple of 4, got {images_u8.dtype} {tuple(images_u8.shape)}")
    x = images_u8.contiguous()
    t = _f32(table256)
    _need_gpu(x, t)
    if t.numel() != 256:
        raise RuntimeError("6dgs_amd: u8_to_planar needs a 256-entry table")
    b, h, w, _ = x.shape
    out = torch.empty(b, 3, h, w, device=x.device)
    check(_lib.load().sixdgs_u8_to_planar(_p(x), b, h * w, _p(t), _p(out), _stream()), "u8_to_planar")
    return out


def image_prep_geometry(h: int, w: int, resize: int = 256, crop: int = 224):
    """The geometry of torchvision's Resize(resize) + CenterCrop(crop) on an h x w image, as BackboneWrapper applies them (backbone.py:52-77):
    -> (resized_h, resized_w, crop_top, crop_left)."""
    if h <= w:
        nh, nw = resize, int(resize * w / h)
    else:
        nh, nw = int(resize * h / w), resize
    return nh, nw, int(round((nh - crop) / 2.0)), int(round((nw - crop) / 2.0))


def image_prep_enabled() -> bool:
    """SIXDGS_IMAGE_PREP=0: uniform RGB batches go through sixdgs_u8_to_planar + PyTorch's antialiased resize, crop and normalisation as before round 6's last step."""
    return os.environ.get("SIXDGS_IMAGE_PREP", "1") != "0"


@_on_device
def image_prep(images_u8: torch.Tensor, table256: torch.Tensor, mean, std, resize: int = 256, crop: int = 224) -> Optional[torch.Tensor]:
    """[B, H, W, 3] uint8 -> [B, 3, crop, crop] fp32 = Normalize(CenterCrop(Resize_bicubic_antialias(table256[value]))) in one launch (sixdgs_image_prep);
    None when the kernel does not take the shape (the caller then uses PyTorch's kernels).  mean / std: three floats each."""
    if images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or images_u8.shape[-1] != 3:
        raise RuntimeError(f"6dgs_amd: image_prep needs uint8 [B, H, W, 3], got {images_u8.dtype} {tuple(images_u8.shape)}")
    x = images_u8.contiguous()
    t = _f32(table256)
    _need_gpu(x, t)
    b, h, w, _ = x.shape
    nh, nw, top, left = image_prep_geometry(h, w, resize, crop)
    if (nh, nw) == (h, w) or nh < crop or nw < crop or t.numel() != 256:
        return None
    m3, s3 = (C.c_float * 3)(*[float(v) for v in mean]), (C.c_float * 3)(*[float(v) for v in std])
    out = torch.empty(b, 3, crop, crop, device=x.device)
    st = _lib.load().sixdgs_image_prep(_p(x), b, h, w, _p(t), nh, nw, top, left, crop, m3, s3, _p(out), _stream())
    if st == -3:                  # SIXDGS_E_UNSUPPORTED: scale / window beyond the kernel's LDS
        return None
    check(st, "image_prep")
    return out


TOK_A_PLAIN, TOK_A_LAYERNORM = 0, 1
TOK_EPI_BIAS, TOK_EPI_GELU, TOK_EPI_RESID = 0, 1, 2


class TokWeights:
    """A dense layer's weight matrix [n, k] as sixdgs_tok_linear wants it: two scaled fp16 planes in the matrix pipe's operand order + reciprocal row
    scales (sixdgs_tok_pack; n a multiple of 128, k of 384).  Built once per weight tensor: `TokWeights.of(w)` keeps the pack on a side table keyed
    by the tensor OBJECT (weak reference), its storage address and its version counter, so an optimiser step, load_state_dict, .to() or a new tensor
    that happens to land on a freed tensor's address re-packs."""
    _cache: dict = {}

    def __init__(self, w: torch.Tensor):
        w = _f32(w)
        _need_gpu(w)
        self.n, self.k = int(w.shape[0]), int(w.shape[1])
        lib = _lib.load()
        self.planes = torch.empty(lib.sixdgs_tok_pack_bytes(self.n, self.k), dtype=torch.uint8, device=w.device)
        self.inv_scale = torch.empty(self.n, dtype=torch.float32, device=w.device)
        check(lib.sixdgs_tok_pack(_p(w), self.n, self.k, w.stride(0), _p(self.planes), _p(self.inv_scale), _stream()), "tok_pack")
        # the pack is kept for the life of the weight and may be read next from ANOTHER stream (a pipeline's image stream, a graph captured elsewhere):
        # finish it here, once per weight (48 x ~25 us for a ViT-S/14), instead of carrying an event through every later call
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(w.device).synchronize()

    @classmethod
    def of(cls, w: torch.Tensor) -> "TokWeights":
        import weakref
        key = id(w)
        hit = cls._cache.get(key)
        if hit is not None and hit[0]() is w and hit[1] == w._version and hit[2] == w.data_ptr():
            return hit[3]
        if w.dim() != 2:
            raise RuntimeError(f"6dgs_amd: a dense layer's weight is [n, k], got {tuple(w.shape)}")
        for k_ in [k_ for k_, v_ in cls._cache.items() if v_[0]() is None]:      # packs of tensors that are gone (an evaluation sweep loads a module per scene)
            del cls._cache[k_]
        with torch.cuda.device(w.device):
            tw = cls(w)
        cls._cache[key] = (weakref.ref(w), w._version, w.data_ptr(), tw)
        return tw


@_on_device
def tok_linear(x: torch.Tensor, w, b: Optional[torch.Tensor] = None, *, ln=None, epilogue: int = TOK_EPI_BIAS, residual: Optional[torch.Tensor] = None,
               gamma: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The backbone stage's dense product with its neighbours folded in (include/sixdgs.h: sixdgs_tok_linear).  w: the layer's weight [n, k] (packed
    on first use, TokWeights.of) or a TokWeights.
      x [M, K] rows (row stride free);  ln=(weight, bias, eps): LayerNorm over K = 384 in front of the product;
      epilogue TOK_EPI_BIAS / _GELU / _RESID (residual [M, N] + gamma [N] * (. + b); out may be the residual tensor)."""
    tw = w if isinstance(w, TokWeights) else TokWeights.of(w)
    x = x.detach()
    if x.dtype != torch.float32:
        x = x.float()
    if x.dim() != 2 or x.shape[1] != tw.k:
        raise RuntimeError(f"6dgs_amd: tok_linear needs x [M, {tw.k}], got {tuple(x.shape)}")
    if x.stride(1) != 1 or x.stride(0) % 4 != 0 or x.data_ptr() % 16 != 0:
        x = x.contiguous()
    _need_gpu(x, tw.planes)
    b = _f32(b) if b is not None else None
    n, k = tw.n, tw.k
    a_mode, ln_w, ln_b, eps = TOK_A_PLAIN, None, None, 0.0
    if ln is not None:
        a_mode, ln_w, ln_b, eps = TOK_A_LAYERNORM, _f32(ln[0]), _f32(ln[1]), float(ln[2])
    m = x.shape[0]
    y = out if out is not None else torch.empty(m, n, device=x.device)
    res = _f32(residual) if residual is not None else None
    check(_lib.load().sixdgs_tok_linear(_p(x), m, k, x.stride(0), a_mode, _p(ln_w), _p(ln_b), eps, _p(tw.planes), _p(tw.inv_scale), _p(b), n, int(epilogue),
                                        _p(res), res.stride(0) if res is not None else 0, _p(_f32(gamma)) if gamma is not None else None, _p(y), y.stride(0),
                                        _stream()), "tok_linear")
    return y


@_on_device
def split_planes_f16(x: torch.Tensor):
    """fp32 [rows,384] -> (scaled fp16 planes as uint8 [rows,1536], reciprocal power-of-two scale of every 128-row tile)."""
    x = _f32(x)
    _need_gpu(x)
    out = torch.empty(x.shape[0], 1536, dtype=torch.uint8, device=x.device)
    scale = torch.empty((x.shape[0] + 127) // 128, device=x.device)
    check(_lib.load().sixdgs_split_planes_f16(_p(x), x.shape[0], x.stride(0), _p(out), _p(scale), _stream()), "split_planes_f16")
    return out, scale


# rays per launch of the ray-MLP chain: 2^20 (6.9 GB of workspace).  A persistent workgroup then walks ~32 tile passes per launch instead of 8, and
# the chain runs 3-4 % faster than with 2^18 (fewer launch tails); any chunking gives the same keys bit for bit.
RAY_KEYS_CHUNK = 1 << 20
RAY_KEYS_CHUNK_MIN = 1 << 16


class Arena:
    """ONE device buffer from which the big per-scene buffers are carved: key planes (1536 B per ray), the select workspace (20 B per ray and image), the
    ray-MLP chain's workspace, a streamed scene's U and chunk planes.  An evaluation sweep walks scenes of very different sizes
    (pretrain_eval_attention.py:224-247: 19 M to 392 M rays in the stand-in sweep): through PyTorch's caching allocator every larger scene meant a new
    100-200 GB hipMalloc (pages mapped and cleared at 30-75 GB/s: seconds) and, because a cached block that small tensors have split can neither be reused
    nor released, an `empty_cache()` between scenes that re-paid it for EVERY scene (round 4: 38 s of set-up against 28 s of evaluation).  The ABI's own
    rule -- the caller provides the workspace -- points at one allocation made once: a bump allocator, reset per scene, with scopes for what lives
    shorter (a chunk of a streamed scene).  Everything carved from it is used on ONE stream (reuse is ordered by the stream); the pipeline's image
    stream never touches it.  Opt-in: `with ops.use_arena(arena): ...` / `ops.set_arena(arena)`; without one the allocations go to torch as before."""

    def __init__(self, nbytes: int, device):
        self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        self.off, self.high = 0, 0
        self._owner = None          # weakref to the ONE module whose per-scene buffers live here (reset(owner) claims it)

    @property
    def capacity(self) -> int:
        return int(self.buf.numel())

    def take(self, nbytes: int, align: int = 4096) -> torch.Tensor:
        base = self.buf.data_ptr()
        off = (base + self.off + align - 1) // align * align - base          # the ADDRESS is aligned (kernels ask for 16 .. 256 B; 4 KB keeps pages apart)
        if off + int(nbytes) > self.buf.numel():
            raise RuntimeError(f"6dgs_amd: arena exhausted ({off + int(nbytes)} of {self.buf.numel()} bytes asked for)")
        self.off = off + int(nbytes)
        self.high = max(self.high, self.off)
        return self.buf[off:self.off]

    def mark(self) -> int:
        return self.off

    def release(self, mark: int) -> None:
        self.off = int(mark)

    def reset(self, owner=None) -> None:
        """Everything carved so far is handed out again.  `owner`: the module doing so -- an arena serves ONE IdentificationModule at a time (its key planes,
        workspaces and sample are dropped by that module before it resets); a reset by a second LIVE module while the first still exists would hand the
        first one's planes out again under its feet, so it is refused.  `release_owner()` passes the arena on."""
        if owner is not None:
            cur = self._owner() if self._owner is not None else None
            if cur is not None and cur is not owner:
                raise RuntimeError("6dgs_amd: this arena already serves another IdentificationModule (ops.Arena serves one module at a time: "
                                   "call arena.release_owner() once the first module's scene is done, or give the second module its own arena)")
            import weakref
            self._owner = weakref.ref(owner)
        self.off = 0

    def release_owner(self) -> None:
        self._owner = None


_arena: Optional[Arena] = None
ARENA_MIN_BYTES = 1 << 22          # smaller buffers stay with torch's allocator


def set_arena(arena: Optional[Arena]) -> Optional[Arena]:
    """Install (or, with None, remove) the arena the big buffers come from; returns the previous one."""
    global _arena
    prev, _arena = _arena, arena
    return prev


def get_arena() -> Optional[Arena]:
    return _arena


class arena_scope:
    """`with ops.arena_scope():` -- what is taken from the arena inside is given back at the end (no-op without an arena)."""

    def __enter__(self):
        self.m = _arena.mark() if _arena is not None else None
        return self

    def __exit__(self, *exc):
        if _arena is not None and self.m is not None:
            _arena.release(self.m)
        return False


def big_empty(shape, dtype, device) -> torch.Tensor:
    """torch.empty for the path's big buffers: from the arena when one is installed on this device (and the buffer is big), else from torch."""
    n = 1
    for d in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)):
        n *= int(d)
    nbytes = n * torch.empty(0, dtype=dtype).element_size()
    a = _arena
    if a is not None and nbytes >= ARENA_MIN_BYTES and a.buf.device == torch.device(device):
        return a.take(nbytes).view(dtype).view(shape)
    return torch.empty(shape, dtype=dtype, device=device)


def _free_bytes(dev) -> int:
    """Device memory a new tensor can still get: what the driver reports free + what PyTorch's caching allocator holds unused."""
    free = torch.cuda.mem_get_info(dev)[0]
    return int(free + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev))


def ray_keys_workspace_bytes(r: int, max_chunk: int = RAY_KEYS_CHUNK) -> int:
    """Transient workspace of one ray_keys call over r rays (callers that budget HBM: bench.py's residency guard, the streamed scorer)."""
    return int(_lib.load().sixdgs_ray_keys_workspace_bytes(int(r), int(max_chunk)))


@_on_device
def ray_keys(ori, dr, rgb, weights: PackedWeights, want_feat: bool = False, want_key: bool = True, max_chunk: int = RAY_KEYS_CHUNK,
             workspace: Optional[torch.Tensor] = None, profile: Optional["KernelProfile"] = None, want_planes: bool = False,
             norm_out: Optional[torch.Tensor] = None):
    """-> (feat | None, key | None)  or, with want_planes (F16X3 modes only), (feat | None, key | None, (uint8 [R,1536] scaled fp16 planes,
    inv_scale [ceil(R/128)])).
    norm_out (device scalar [1], scaled fp16 planes only): updated to max(norm_out, max_r |key row|) -- see key_norm_max."""
    ori, dr, rgb = _f32(ori), _f32(dr), _f32(rgb)
    _need_gpu(ori, dr, rgb)
    lib = _lib.load()
    r = ori.shape[0]
    dev = ori.device
    feat = torch.empty(r, D, device=dev) if want_feat else None
    key = torch.empty(r, D, device=dev) if want_key else None
    mode = effective_mma_mode()
    f16 = want_planes and mode in F16_MODES
    if want_planes and not f16:
        raise RuntimeError("6dgs_amd: key planes exist in the fp16 x 3 modes only (MMA_F32 / MMA_BF16X6 score on fp32 keys)")
    planes = big_empty((r, 1536), torch.uint8, dev) if want_planes else None
    inv = torch.empty((r + 127) // 128, device=dev) if f16 else None
    nbytes = lib.sixdgs_ray_keys_workspace_bytes(r, int(max_chunk))
    if workspace is None or workspace.numel() < nbytes:
        # 6560 B of transient workspace per ray of a chunk (6.9 GB at the default 2^20 rays): next to a scene that nearly fills the HBM the
        # chunk shrinks instead of the allocation failing -- any chunking gives the same keys bit for bit, smaller chunks are 3 % slower
        chunk = int(max_chunk)
        room = (_arena.capacity - _arena.off) if (_arena is not None and _arena.buf.device == dev) else 0.5 * _free_bytes(dev)
        while chunk > RAY_KEYS_CHUNK_MIN and nbytes > room:
            chunk //= 2
            nbytes = lib.sixdgs_ray_keys_workspace_bytes(r, chunk)
    if norm_out is not None and not f16:
        raise RuntimeError("6dgs_amd: norm_out needs scaled fp16 key planes (want_planes in MMA_F16X3 mode)")
    with arena_scope():         # the chain's workspace is transient: back to the arena when the call is enqueued (reuse is ordered by the stream)
        ws = workspace if workspace is not None and workspace.numel() >= nbytes else big_empty(nbytes, torch.uint8, dev)
        check(lib.sixdgs_ray_keys_ex(_p(ori), _p(dr), _p(rgb), r, weights.ref, _p(feat), _p(key), _p(planes), _p(inv), _p(norm_out), _p(ws), ws.numel(),
                                     _stream(), profile.ref if profile is not None else None, mode), "ray_keys")
    if want_planes:
        return feat, key, (planes, inv)
    return feat, key


@_on_device
def pad_tokens(token_list, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """list of [T_i, 398] -> tokens [B,256,398] (zero padded), n_tok int32 [B] (device)."""
    b = len(token_list)
    tok = torch.zeros(b, MAX_TOKENS, TOK_IN, device=device)
    n = torch.empty(b, dtype=torch.int32)
    for i, t in enumerate(token_list):
        if t.shape[0] > MAX_TOKENS or t.shape[1] != TOK_IN:
            raise RuntimeError(f"6dgs_amd: token block {tuple(t.shape)} does not fit [<=256, 398]")
        tok[i, : t.shape[0]] = t
        n[i] = t.shape[0]
    return tok, n.to(device)


@_on_device
def q_proj(tokens: torch.Tensor, n_tok: torch.Tensor, weights: PackedWeights) -> torch.Tensor:
    tokens = _f32(tokens)
    _need_gpu(tokens, n_tok)
    b = tokens.shape[0]
    q = torch.empty(b, MAX_TOKENS, D, device=tokens.device)
    check(_lib.load().sixdgs_q_proj(_p(tokens), _p(n_tok), b, weights.ref, _p(q), _stream()), "q_proj")
    return q


def score_topk_workspace_bytes(r: int, batch: int, topk: int = 100, planes: bool = False) -> int:
    """Workspace of the scorer for `batch` resident images.  planes=True: the call will run on pre-split key planes in the
    active MMA mode (24-bit logits in MMA_F16X3: 23 % smaller); False: enough for every mode."""
    if planes:
        return int(_lib.load().sixdgs_score_topk_workspace_bytes_ex(int(r), int(batch), int(topk), effective_mma_mode(), 1))
    return int(_lib.load().sixdgs_score_topk_workspace_bytes(int(r), int(batch), int(topk)))


@_on_device
def score_topk(q: torch.Tensor, n_tok: torch.Tensor, key: Optional[torch.Tensor], topk: int = 100, want_scores: bool = True,
               want_stats: bool = False, workspace: Optional[torch.Tensor] = None, images_in_flight: Optional[int] = None,
               profile: Optional["KernelProfile"] = None, n_tok_host=None, key_planes: Optional[torch.Tensor] = None,
               key_scale: Optional[torch.Tensor] = None):
    """key: fp32 [R,384] and/or key_planes: uint8 [R,1536] scaled fp16 planes + key_scale (the F16X3 modes: the DMA-fed kernels);
    MMA_F32 / MMA_BF16X6 score on the fp32 keys."""
    q = _f32(q)
    key = _f32(key) if key is not None else None
    _need_gpu(q, key, n_tok, key_planes)
    lib = _lib.load()
    r = key.shape[0] if key is not None else key_planes.shape[0]
    b, dev = q.shape[0], q.device
    idx = torch.empty(b, topk, dtype=torch.int64, device=dev)
    val = torch.empty(b, topk, device=dev)
    scores = torch.empty(b, r, device=dev) if want_scores else None
    stats = torch.empty(b, MAX_TOKENS, 2, device=dev) if want_stats else None
    if workspace is None:
        inflight = b if images_in_flight is None else max(1, min(b, images_in_flight))
        workspace = torch.empty(score_topk_workspace_bytes(r, inflight, topk, planes=key_planes is not None), dtype=torch.uint8, device=dev)
    h_n = None
    if profile is not None and n_tok_host is not None:
        h_n = (C.c_int32 * b)(*[int(v) for v in n_tok_host])
    mode = effective_mma_mode()
    if key_planes is not None and mode in F16_MODES:
        if key_planes.shape[1] != 1536 or key_scale is None:
            raise RuntimeError("6dgs_amd: key planes are not in the format of the active MMA mode (scaled fp16 planes [R,1536] + key_scale)")
    elif key_planes is not None:
        if key is None:
            raise RuntimeError("6dgs_amd: MMA_F32 / MMA_BF16X6 score on fp32 keys: pass `key` (key planes are the operand of the fp16 x 3 kernels only)")
        key_planes = key_scale = None
    check(lib.sixdgs_score_topk_ex(_p(q), _p(n_tok), h_n, b, _p(key), _p(key_planes), _p(key_scale), r, int(topk), _p(scores), _p(idx), _p(val), _p(stats),
                                   _p(workspace), workspace.numel(), _stream(), profile.ref if profile is not None else None,
                                   _mma_mode), "score_topk")
    return idx, val, scores, stats


SELECT_MAX_CANDIDATES = 4096     # candidates re-scored exactly per image; an image with more falls back to the two-pass scorer
SELECT_SAMPLE_STRIDE = 16        # the pre-pass sees one ray in 16 ...
SELECT_SAMPLE_STRIDE_LARGE = 32  # ... one in 32 from SELECT_LARGE_RAYS rays (the sample is then still >= 0.5 M rays: its row sums are
SELECT_LARGE_RAYS = 16_000_000   # as representative as one in 16 of a scene half the size, and the pre-pass costs half) ...
SELECT_SAMPLE_STRIDE_HUGE = 64   # ... one in 64 from SELECT_HUGE_RAYS rays (>= 0.5 M sample rays again).  The sample only sets the
SELECT_HUGE_RAYS = 32_000_000    # exponent offsets and the width g_min / g_max of the bounds, never the answer
SELECT_MIN_RAYS = 1 << 20        # below this the two-pass scorer is as fast (its logits fit in cache-sized workspaces)
_select_enabled = True


def set_select_enabled(on: bool):
    """Whether top-k-only requests on fp16 key planes may take sixdgs_score_select (no logits through HBM)."""
    global _select_enabled
    _select_enabled = bool(on)


def select_enabled() -> bool:
    return _select_enabled


@_on_device
def select_sample_indices(r: int, device, stride: Optional[int] = None) -> torch.Tensor:
    """One ray of every `stride` consecutive ones (default: 16, 32 from SELECT_LARGE_RAYS rays, 64 from SELECT_HUGE_RAYS), at a position that varies
    pseudo-randomly from group to group (a fixed position would pick the same iso-cell direction of every ellipsoid):
    indices stride*i + (2654435761 i mod 2^32 >> 13) mod stride."""
    if stride is None:
        stride = SELECT_SAMPLE_STRIDE_HUGE if r >= SELECT_HUGE_RAYS else (SELECT_SAMPLE_STRIDE_LARGE if r >= SELECT_LARGE_RAYS else SELECT_SAMPLE_STRIDE)
    n = r // stride
    i = torch.arange(n, dtype=torch.int64, device=device)
    return i * stride + (((i * 2654435761) & 0xFFFFFFFF) >> 13) % stride


@_on_device
def key_norm_max(planes: torch.Tensor, scale: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """max_r |k_r| of the rows of scaled fp16 key planes as a device scalar [1] (rounded up: the select path's error bound is built
    on it, include/sixdgs.h).  `out`: accumulate into an existing scalar (scenes that go through in chunks)."""
    _need_gpu(planes, scale, out)
    if planes.shape[1] != 1536:
        raise RuntimeError("6dgs_amd: key_norm_max needs scaled fp16 key planes (MMA_F16X3)")
    if out is None:
        out = torch.zeros(1, device=planes.device)
    check(_lib.load().sixdgs_key_planes_norm_max(_p(planes), _p(scale), planes.shape[0], _p(out), _stream()), "key_planes_norm_max")
    return out


def score_select_workspace_bytes(r: int, batch: int, topk: int = 100, max_candidates: int = SELECT_MAX_CANDIDATES) -> int:
    return int(_lib.load().sixdgs_score_select_workspace_bytes(int(r), int(batch), int(topk), int(max_candidates)))


def select_sweep_plan(n_tok_host, batch: Optional[int] = None):
    """How the select sweep cuts a batch with these token counts (None: unknown, one image per 256-token tile) into launches:
    -> list of (tiles, images) per launch (include/sixdgs.h: sixdgs_select_sweep_plan; host arithmetic, no GPU needed)."""
    lib = _lib.load()
    b = len(n_tok_host) if n_tok_host is not None else int(batch)
    h_n = (C.c_int32 * b)(*[int(v) for v in n_tok_host]) if n_tok_host is not None else None
    cap = max(1, b)
    slots, imgs = (C.c_int32 * cap)(), (C.c_int32 * cap)()
    n = lib.sixdgs_select_sweep_plan(h_n, b, slots, imgs, cap)
    if n < 0:
        check(n, "select_sweep_plan")
    return [(int(slots[i]), int(imgs[i])) for i in range(min(n, cap))]


@_on_device
def score_select(q: torch.Tensor, n_tok: torch.Tensor, key_planes: torch.Tensor, key_scale: torch.Tensor, sample_planes: torch.Tensor,
                 sample_scale: torch.Tensor, topk: int = 100, max_candidates: int = SELECT_MAX_CANDIDATES,
                 workspace: Optional[torch.Tensor] = None, profile: Optional["KernelProfile"] = None, n_tok_host=None,
                 key_norm: Optional[torch.Tensor] = None):
    """Top-k without materialised logits (include/sixdgs.h: sixdgs_score_select).  Returns (idx [B,k], val [B,k], status [B] int32 on
    the device: candidates examined, or -1 = this image needs the two-pass scorer).  key_norm: key_norm_max(key_planes, key_scale),
    computed here (one more pass over the planes) when not handed in -- callers with a key cache keep it beside the planes.
    n_tok_host: list of the token counts (must equal n_tok): masked views then share the sweep's 256-token tiles (token packing)."""
    q = _f32(q)
    _need_gpu(q, n_tok, key_planes, key_scale, sample_planes, sample_scale, key_norm)
    if key_norm is None:
        key_norm = key_norm_max(key_planes, key_scale)
    if key_planes.shape[1] != 1536 or sample_planes.shape[1] != 1536:
        raise RuntimeError("6dgs_amd: score_select needs scaled fp16 key planes (MMA_F16X3)")
    lib = _lib.load()
    b, dev, r, rs = q.shape[0], q.device, key_planes.shape[0], sample_planes.shape[0]
    idx = torch.empty(b, topk, dtype=torch.int64, device=dev)
    val = torch.empty(b, topk, device=dev)
    status = torch.empty(b, dtype=torch.int32, device=dev)
    if workspace is None:
        workspace = torch.empty(score_select_workspace_bytes(r, b, topk, max_candidates), dtype=torch.uint8, device=dev)
    # the host copy of the token counts: with it the sweep PACKS the images of a launch by their token counts (round 5; include/sixdgs.h)
    h_n = (C.c_int32 * b)(*[int(v) for v in n_tok_host]) if n_tok_host is not None else None
    check(lib.sixdgs_score_select(_p(q), _p(n_tok), h_n, b, _p(key_planes), _p(key_scale), _p(key_norm), r, _p(sample_planes), _p(sample_scale), rs,
                                  int(topk), int(max_candidates), _p(idx), _p(val), _p(status), _p(workspace), workspace.numel(), _stream(),
                                  profile.ref if profile is not None else None), "score_select")
    return idx, val, status


class SelectStream:
    """The select path stage by stage (include/sixdgs.h: sixdgs_select_begin / _sweep / _candidates / _rescore) for ray sets that
    go through in chunks: begin(sample planes) -> sweep(chunk planes, ray offset) per chunk -> candidates() -> rescore(planes of
    the candidates).  Holds the caller-side buffers: ctok, gsum [B,256], U [B, R rounded up to 256]."""

    @_on_device
    def __init__(self, q: torch.Tensor, n_tok: torch.Tensor, r_total: int, topk: int = 100, max_candidates: int = SELECT_MAX_CANDIDATES,
                 n_tok_host=None):
        self.q, self.n_tok = _f32(q), n_tok
        _need_gpu(self.q, n_tok)
        self.b, self.dev = self.q.shape[0], self.q.device
        self.device = self.dev
        self.r, self.topk, self.cmax = int(r_total), int(topk), int(max_candidates)
        self.stride = (self.r + 255) // 256 * 256
        self.ctok = torch.empty(self.b, MAX_TOKENS, device=self.dev)
        self.gsum = torch.empty(self.b, MAX_TOKENS, device=self.dev)
        self.u = big_empty((self.b, self.stride), torch.float32, self.dev)
        self.utm = torch.empty(self.b, self.stride // 256, device=self.dev)        # the largest U of every 256-ray tile (candidate threshold)
        self.key_norm = torch.zeros(1, device=self.dev)        # max |k_r| over the chunks swept so far (the bound of the candidate stage)
        self.ws = torch.empty(1, dtype=torch.uint8, device=self.dev)
        self.h_n = (C.c_int32 * self.b)(*[int(v) for v in n_tok_host]) if n_tok_host is not None else None

    def _grow(self, need):
        if self.ws.numel() < need:
            self.ws = None
            self.ws = big_empty(need, torch.uint8, self.dev)       # (from an arena: the smaller predecessor stays taken until the caller's scope ends)

    def reserve(self, chunk_rows: int):
        """Grow the stage workspace ONCE to what sweep(chunk of `chunk_rows` rays), candidates() and rescore() will ask for -- callers that take the
        chunks' planes from an arena inside a per-chunk scope must not have the workspace grow (and be given back) inside such a scope."""
        lib = _lib.load()
        self._grow(max(lib.sixdgs_select_workspace_bytes(int(chunk_rows), self.b, self.topk, self.cmax),
                       lib.sixdgs_select_candidates_workspace_bytes(self.r, self.b, self.topk, self.cmax)))

    @_on_device
    def begin(self, sample_planes, sample_scale):
        _need_gpu(self.q, sample_planes, sample_scale)
        lib = _lib.load()
        self._grow(lib.sixdgs_select_workspace_bytes(sample_planes.shape[0], self.b, self.topk, self.cmax))
        check(lib.sixdgs_select_begin(_p(self.q), _p(self.n_tok), self.h_n, self.b, _p(sample_planes), _p(sample_scale), sample_planes.shape[0], self.r,
                                      _p(self.ctok), _p(self.gsum), _p(self.ws), self.ws.numel(), _stream()), "select_begin")

    @_on_device
    def sample_stats(self, sample_planes, sample_scale) -> torch.Tensor:
        """First half of begin(): (max, sumexp) [B,256,2] of the given sample -- ray-sharded scenes merge them across the shards."""
        _need_gpu(self.q, sample_planes, sample_scale)
        lib = _lib.load()
        self._grow(lib.sixdgs_select_workspace_bytes(sample_planes.shape[0], self.b, self.topk, self.cmax))
        stats = torch.empty(self.b, MAX_TOKENS, 2, device=self.dev)
        check(lib.sixdgs_select_sample_stats(_p(self.q), _p(self.n_tok), self.h_n, self.b, _p(sample_planes), _p(sample_scale), sample_planes.shape[0],
                                             _p(stats), _p(self.ws), self.ws.numel(), _stream()), "select_sample_stats")
        return stats

    @_on_device
    def prepare(self, stats: torch.Tensor, r_sample_total: int, r_total: int):
        """Second half of begin(): the (merged) sample statistics and the TOTAL sample / ray counts of the scene -> ctok; gsum = 0."""
        stats = _f32(stats)
        _need_gpu(self.q, stats)
        check(_lib.load().sixdgs_select_prepare(_p(stats), _p(self.n_tok), self.b, int(r_sample_total), int(r_total), _p(self.ctok), _p(self.gsum),
                                                _stream()), "select_prepare")

    @_on_device
    def topk_u(self, exact: bool = False) -> torch.Tensor:
        """The k largest TILE MAXIMA of U over the rays swept so far, per image, descending (the k largest U themselves for scenes of fewer than
        2 k tiles): the k-th of them is a lower bound of the k-th largest U, which is what the candidate threshold needs.  exact=True: the
        k largest U themselves (six passes over U instead of one over the tile maxima) -- the bound that does not degrade when the top
        rays sit in a few tiles."""
        lib = _lib.load()
        _need_gpu(self.q)
        self._grow(lib.sixdgs_select_candidates_workspace_bytes(self.r, self.b, self.topk, self.cmax))
        val = torch.empty(self.b, self.topk, device=self.dev)
        check(lib.sixdgs_select_topk_u(_p(self.u), self.stride, self.r, None if exact else _p(self.utm), self.b, self.topk, _p(val), _p(self.ws),
                                       self.ws.numel(), _stream()),
              "select_topk_u")
        return val

    @_on_device
    def sweep(self, planes, scale, ray_offset: int, profile: Optional["KernelProfile"] = None, update_norm: bool = True):
        """update_norm=False: the caller already folded these planes' largest key norm into self.key_norm (ray_keys(norm_out=...))."""
        if ray_offset % 256:
            raise RuntimeError("6dgs_amd: select sweep chunks must start at a multiple of 256 rays")
        _need_gpu(self.q, planes, scale)
        lib = _lib.load()
        rc = planes.shape[0]
        self._grow(lib.sixdgs_select_workspace_bytes(rc, self.b, self.topk, self.cmax))
        check(lib.sixdgs_select_sweep(_p(self.q), _p(self.n_tok), self.h_n, self.b, _p(planes), _p(scale), rc,
                                      _p(self.ctok), _p(self.gsum), C.c_void_p(self.u.data_ptr() + 4 * int(ray_offset)), self.stride,
                                      C.c_void_p(self.utm.data_ptr() + 4 * (int(ray_offset) // 256)),
                                      _p(self.ws), self.ws.numel(), _stream(), profile.ref if profile is not None else None), "select_sweep")
        if update_norm:
            key_norm_max(planes, scale, out=self.key_norm)

    @_on_device
    def candidates(self, uk: Optional[torch.Tensor] = None):
        """-> (cand [B,cmax] int64 ascending ray indices, count [B] int32 on the device).  uk [B]: the k-th largest U of the WHOLE
        scene (ray-sharded callers, from the merged topk_u lists); default: of this object's rays."""
        lib = _lib.load()
        _need_gpu(self.q)
        self._grow(lib.sixdgs_select_candidates_workspace_bytes(self.r, self.b, self.topk, self.cmax))
        cand = torch.zeros(self.b, self.cmax, dtype=torch.int64, device=self.dev)
        count = torch.empty(self.b, dtype=torch.int32, device=self.dev)
        uk = _f32(uk) if uk is not None else None
        check(lib.sixdgs_select_candidates(_p(self.u), self.stride, self.r, _p(self.utm), _p(self.q), _p(self.n_tok), self.b, _p(self.gsum), _p(self.key_norm),
                                           _p(uk), self.topk, self.cmax, _p(cand), _p(count), _p(self.ws), self.ws.numel(), _stream()), "select_candidates")
        return cand, count

    @_on_device
    def rescore(self, planes, scale, cand, count, compact: bool, allow_fewer: bool = False):
        """-> (idx [B,k], val [B,k], status [B] int32).  allow_fewer: fewer than k candidates is an answer (-1 / NaN padded), not a
        refusal -- a shard of a ray-sharded scene."""
        lib = _lib.load()
        _need_gpu(self.q, planes, scale, cand, count)
        idx = torch.empty(self.b, self.topk, dtype=torch.int64, device=self.dev)
        val = torch.empty(self.b, self.topk, device=self.dev)
        status = torch.empty(self.b, dtype=torch.int32, device=self.dev)
        self._grow(lib.sixdgs_select_candidates_workspace_bytes(self.r, self.b, self.topk, self.cmax))
        check(lib.sixdgs_select_rescore(_p(self.q), _p(self.n_tok), self.b, _p(planes), _p(scale), 1 if compact else 0, _p(self.ctok),
                                        _p(self.gsum), _p(cand), _p(count), self.r, self.topk, self.cmax, 1 if allow_fewer else 0, _p(idx), _p(val), _p(status),
                                        _p(self.ws), self.ws.numel(), _stream()), "select_rescore")
        return idx, val, status


@_on_device
def score_pass1(q: torch.Tensor, n_tok: torch.Tensor, key: Optional[torch.Tensor], workspace: torch.Tensor, topk: int = 100,
                key_planes: Optional[torch.Tensor] = None, key_scale: Optional[torch.Tensor] = None,
                profile: Optional["KernelProfile"] = None, n_tok_host=None) -> torch.Tensor:
    """First half of the ray-sharded scorer (include/sixdgs.h: sixdgs_score_pass1): logits of this shard for all images
    stay in `workspace` (score_topk_workspace_bytes(r, batch, topk) bytes); returns the shard's row statistics [B,256,2]."""
    q = _f32(q)
    key = _f32(key) if key is not None else None
    _need_gpu(q, key, n_tok, key_planes, workspace)
    r = key.shape[0] if key is not None else key_planes.shape[0]
    b = q.shape[0]
    stats = torch.empty(b, MAX_TOKENS, 2, device=q.device)
    h_n = (C.c_int32 * b)(*[int(v) for v in n_tok_host]) if (profile is not None and n_tok_host is not None) else None
    check(_lib.load().sixdgs_score_pass1(_p(q), _p(n_tok), h_n, b, _p(key), _p(key_planes), _p(key_scale), r, int(topk), _p(stats),
                                         _p(workspace), workspace.numel(), _stream(), profile.ref if profile is not None else None,
                                         _mma_mode), "score_pass1")
    return stats


@_on_device
def score_pass2(stats: torch.Tensor, n_tok: torch.Tensor, r: int, workspace: torch.Tensor, topk: int = 100, used_planes: bool = True,
                want_scores: bool = True):
    """Second half: global row statistics [B,256,2] -> (local idx [B,k], val [B,k], scores [B,r] | None) of this shard."""
    stats = _f32(stats)
    _need_gpu(stats, n_tok, workspace)
    b, dev = stats.shape[0], stats.device
    idx = torch.empty(b, topk, dtype=torch.int64, device=dev)
    val = torch.empty(b, topk, device=dev)
    scores = torch.empty(b, r, device=dev) if want_scores else None
    check(_lib.load().sixdgs_score_pass2(_p(stats), _p(n_tok), b, 1 if used_planes else 0, int(r), int(topk), _p(scores), _p(idx), _p(val),
                                         _p(workspace), workspace.numel(), _stream(), _mma_mode), "score_pass2")
    return idx, val, scores


@_on_device
def topk(scores: torch.Tensor, k: int = 100):
    scores = _f32(scores)
    _need_gpu(scores)
    lib = _lib.load()
    s2 = scores.reshape(-1, scores.shape[-1])
    b, r = s2.shape
    idx = torch.empty(b, k, dtype=torch.int64, device=s2.device)
    val = torch.empty(b, k, device=s2.device)
    ws = torch.empty(lib.sixdgs_topk_workspace_bytes(r, b, k), dtype=torch.uint8, device=s2.device)
    check(lib.sixdgs_topk(_p(s2), r, b, int(k), _p(idx), _p(val), _p(ws), ws.numel(), _stream()), "topk")
    return idx.reshape(*scores.shape[:-1], k), val.reshape(*scores.shape[:-1], k)


# ---------------------------------------------------------------------------------------------
# DistanceBasedScoreLoss target scores
# ---------------------------------------------------------------------------------------------
@_on_device
def distance_target(rays_ori, rays_dir, pose, n_tokens: int, want_sum: bool = False):
    """Target scores [R] of distance_based_loss.py for the ground-truth c2w `pose` [4,4] (any device), summing to n_tokens."""
    rays_ori, rays_dir = _f32(rays_ori), _f32(rays_dir)
    _need_gpu(rays_ori, rays_dir)
    lib = _lib.load()
    r, dev = rays_ori.shape[0], rays_ori.device
    pose = _f32(pose.to(dev)).reshape(16)
    target = torch.empty(r, device=dev)
    tot = torch.zeros(1, device=dev) if want_sum else None
    ws = torch.empty(lib.sixdgs_distance_target_workspace_bytes(r), dtype=torch.uint8, device=dev)
    check(lib.sixdgs_distance_target(_p(rays_ori), _p(rays_dir), r, _p(pose), int(n_tokens), _p(target), _p(tot), _p(ws), ws.numel(),
                                     _stream()), "distance_target")
    return (target, tot) if want_sum else target


# ---------------------------------------------------------------------------------------------
# pose
# ---------------------------------------------------------------------------------------------
@_on_device
def solve_pose(rays_ori, rays_dir, idx, val, up, gt_c2w=None):
    """Batched pose tail.  idx/val [B,k], up [B,3], gt_c2w [B,4,4] or None.
    Returns dict(c2w[B,4,4], status[B], w_final[B,k], n_kept[B], centre[B,3], errors[B,2])."""
    rays_ori, rays_dir, val, up = _f32(rays_ori), _f32(rays_dir), _f32(val), _f32(up)
    idx = _i64(idx)
    _need_gpu(rays_ori, rays_dir, idx, val, up)
    b, k = idx.shape
    dev = idx.device
    gt = _f32(gt_c2w) if gt_c2w is not None else None
    c2w = torch.empty(b, 4, 4, device=dev)
    status = torch.empty(b, dtype=torch.int32, device=dev)
    wf = torch.empty(b, k, device=dev)
    nk = torch.empty(b, dtype=torch.int32, device=dev)
    ctr = torch.empty(b, 3, device=dev)
    err = torch.empty(b, 2, device=dev)
    check(_lib.load().sixdgs_solve_pose(_p(rays_ori), _p(rays_dir), rays_ori.shape[0], _p(idx), _p(val), k, _p(up), _p(gt), b,
                                        _p(c2w), _p(status), _p(wf), _p(nk), _p(ctr), _p(err), _stream()), "solve_pose")
    return dict(c2w=c2w, status=status, w_final=wf, n_kept=nk, centre=ctr, errors=err)
