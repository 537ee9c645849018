"""ctypes binding of lib6dgs_hip.so (the C ABI declared in include/sixdgs.h).

The product path has NO fallback: if the shared library is missing or fails to load, importing an
op raises RuntimeError.  PyTorch is imported first so that the library's libamdhip64.so.7 dependency
resolves to the HIP runtime PyTorch already loaded (one runtime per process: streams and device
pointers are shared between torch and the kernels).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "lib6dgs_hip.so")
if os.environ.get("SIXDGS_LIB"):       # developer switch: a private build of the library (tools/build_variant.py); never set in production
    LIB_PATH = os.path.abspath(os.environ["SIXDGS_LIB"])

vp = C.c_void_p
i64 = C.c_int64
i32 = C.c_int
sz = C.c_size_t


PROFILE_SLOTS = 128
ABI_VERSION = 7          # must equal SIXDGS_ABI_VERSION in include/sixdgs.h (checked by __graft_entry__.post_build_checks)


class Profile(C.Structure):
    _fields_ = [("count", i32), ("start", vp * PROFILE_SLOTS), ("stop", vp * PROFILE_SLOTS),
                ("flops", C.c_double * PROFILE_SLOTS), ("bytes", C.c_double * PROFILE_SLOTS)]


class ScorerWeights(C.Structure):
    _fields_ = [(n, vp) for n in ("w1", "b1", "w2", "b2", "w3", "b3", "w4", "b4", "wk", "bk", "wq", "bq", "m1", "m2", "m3", "m4", "mk", "w4k", "b4k", "m4k", "planes")]


# name -> (restype, argtypes); mirrors include/sixdgs.h one to one
SIGNATURES = {
    "sixdgs_abi_version": (i32, []),
    "sixdgs_error_string": (C.c_char_p, [i32]),
    "sixdgs_mask_degraded": (i32, [vp, i64, i32, vp, vp]),
    "sixdgs_sym_eig_3x3": (i32, [vp, i64, vp, vp, vp]),
    "sixdgs_normals_knn": (i32, [vp, i64, vp, i64, i32, vp, vp, vp]),
    "sixdgs_normals_knn_grid_workspace_bytes": (sz, [i64]),
    "sixdgs_normals_knn_grid": (i32, [vp, i64, vp, i64, i32, vp, vp, vp, sz, vp]),
    "sixdgs_emit_quadricell_count": (i32, [vp, vp, i32, vp, vp, i64, vp, i32, i32, vp, vp, vp, vp]),
    "sixdgs_emit_quadricell_write": (i32, [vp, vp, i32, vp, vp, vp, i32, i32, vp, i64, vp, i32, i32, vp, vp, vp, vp, vp, vp]),
    "sixdgs_quadricell_cell_counts": (i32, [vp, i64, i32, vp, vp, vp, vp]),
    "sixdgs_quadricell_centers": (i32, [vp, i64, i32, i32, vp, vp, vp, vp]),
    "sixdgs_isocell_distribution": (i32, [i32, i32, vp, C.POINTER(i64), vp]),
    "sixdgs_rotate_isocell": (i32, [vp, i64, vp, i64, vp, vp]),
    "sixdgs_emit_isocell": (i32, [vp, vp, i32, vp, vp, vp, i32, i32, vp, i64, vp, vp, i64, vp, vp, vp, vp, vp]),
    "sixdgs_eval_sh_color": (i32, [vp, i32, vp, i64, i32, vp, vp]),
    "sixdgs_packed_weights_floats": (sz, []),
    "sixdgs_pack_weights": (i32, [vp] * 13 + [C.POINTER(ScorerWeights), vp]),
    "sixdgs_ray_encode": (i32, [vp, vp, vp, i64, vp, vp]),
    "sixdgs_ray_keys_workspace_bytes": (sz, [i64, i64]),
    "sixdgs_ray_keys": (i32, [vp, vp, vp, i64, C.POINTER(ScorerWeights), vp, vp, vp, sz, vp]),
    "sixdgs_ray_keys_ex": (i32, [vp, vp, vp, i64, C.POINTER(ScorerWeights), vp, vp, vp, vp, vp, vp, sz, vp, C.POINTER(Profile), i32]),
    "sixdgs_key_planes_f16_bytes": (sz, [i64]),
    "sixdgs_split_planes_f16": (i32, [vp, i64, i64, vp, vp, vp]),
    "sixdgs_score_pass1": (i32, [vp, vp, vp, i32, vp, vp, vp, i64, i32, vp, vp, sz, vp, C.POINTER(Profile), i32]),
    "sixdgs_score_pass2": (i32, [vp, vp, i32, i32, i64, i32, vp, vp, vp, vp, sz, vp, i32]),
    "sixdgs_score_topk_ex": (i32, [vp, vp, vp, i32, vp, vp, vp, i64, i32, vp, vp, vp, vp, vp, sz, vp, C.POINTER(Profile), i32]),
    "sixdgs_score_select_workspace_bytes": (sz, [i64, i32, i32, i32]),
    "sixdgs_score_select": (i32, [vp, vp, vp, i32, vp, vp, vp, i64, vp, vp, i64, i32, i32, vp, vp, vp, vp, sz, vp, C.POINTER(Profile)]),
    "sixdgs_tok_attention": (i32, [vp, i64, i32, i32, i32, vp, i64, vp]),
    "sixdgs_im2col": (i32, [vp, i64, i64, i64, i64, i32, i32, i32, i32, i32, i32, vp, vp]),
    "sixdgs_u8_to_planar": (i32, [vp, i32, i64, vp, vp, vp]),
    "sixdgs_image_prep": (i32, [vp, i32, i32, i32, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]),
    "sixdgs_tok_pack_bytes": (sz, [i32, i32]),
    "sixdgs_tok_pack": (i32, [vp, i32, i32, i64, vp, vp, vp]),
    "sixdgs_tok_linear": (i32, [vp, i64, i32, i64, i32, vp, vp, C.c_float, vp, vp, vp, i32, i32, vp, i64, vp, vp, i64, vp]),
    "sixdgs_key_planes_norm_max": (i32, [vp, vp, i64, vp, vp]),
    "sixdgs_select_workspace_bytes": (sz, [i64, i32, i32, i32]),
    "sixdgs_select_candidates_workspace_bytes": (sz, [i64, i32, i32, i32]),
    "sixdgs_select_begin": (i32, [vp, vp, vp, i32, vp, vp, i64, i64, vp, vp, vp, sz, vp]),
    "sixdgs_select_sweep_plan": (i32, [vp, i32, vp, vp, i32]),
    "sixdgs_select_sweep": (i32, [vp, vp, vp, i32, vp, vp, i64, vp, vp, vp, i64, vp, vp, sz, vp, C.POINTER(Profile)]),
    "sixdgs_select_candidates": (i32, [vp, i64, i64, vp, vp, vp, i32, vp, vp, vp, i32, i32, vp, vp, vp, sz, vp]),
    "sixdgs_select_rescore": (i32, [vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, i64, i32, i32, i32, vp, vp, vp, vp, sz, vp]),
    "sixdgs_select_sample_stats": (i32, [vp, vp, vp, i32, vp, vp, i64, vp, vp, sz, vp]),
    "sixdgs_select_prepare": (i32, [vp, vp, i32, i64, i64, vp, vp, vp]),
    "sixdgs_select_topk_u": (i32, [vp, i64, i64, vp, i32, i32, vp, vp, sz, vp]),
    "sixdgs_linear_splitk_workspace_bytes": (sz, [i64, i32, i32]),
    "sixdgs_linear_splitk": (i32, [vp, i64, i32, i64, vp, i64, vp, i32, i32, vp, i64, i32, vp, sz, vp, i32]),
    "sixdgs_linear_ex": (i32, [vp, i64, i32, i64, vp, i64, vp, i32, i32, vp, i64, vp, i32]),
    "sixdgs_profile_collect": (i32, [C.POINTER(Profile), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.POINTER(i32)]),
    "sixdgs_linear": (i32, [vp, i64, i32, i64, vp, i64, vp, i32, i32, vp, i64, vp]),
    "sixdgs_q_proj": (i32, [vp, vp, i32, C.POINTER(ScorerWeights), vp, vp]),
    "sixdgs_score_topk_workspace_bytes": (sz, [i64, i32, i32]),
    "sixdgs_score_topk_workspace_bytes_ex": (sz, [i64, i32, i32, i32, i32]),
    "sixdgs_score_topk": (i32, [vp, vp, i32, vp, i64, i32, vp, vp, vp, vp, vp, sz, vp]),
    "sixdgs_topk_workspace_bytes": (sz, [i64, i32, i32]),
    "sixdgs_topk": (i32, [vp, i64, i32, i32, vp, vp, vp, sz, vp]),
    "sixdgs_distance_target_workspace_bytes": (sz, [i64]),
    "sixdgs_distance_target": (i32, [vp, vp, i64, vp, i32, vp, vp, vp, sz, vp]),
    "sixdgs_solve_pose": (i32, [vp, vp, i64, vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]),
}

_lib = None


def load():
    """Load the library and attach the prototypes.  Raises RuntimeError when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (must come first: shares its HIP runtime with the kernels)

    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"6dgs_amd: {LIB_PATH} is missing -- build it with `python -m 6dgs_amd.build` / __graft_entry__.build(); "
            "there is no CPU fallback for the product path")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise RuntimeError(f"6dgs_amd: cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"6dgs_amd: {LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.sixdgs_abi_version() != ABI_VERSION:
        raise RuntimeError(f"6dgs_amd: ABI version mismatch (library {lib.sixdgs_abi_version()}, binding {ABI_VERSION})")
    _lib = lib
    return lib


def check(status: int, what: str = ""):
    if status != 0:
        msg = load().sixdgs_error_string(int(status)).decode()
        raise RuntimeError(f"6dgs_amd: {what or 'call'} failed: {msg} (status {status})")
