"""Where the wall time of the per-scene key set-up goes (2.7 s wall for 0.46 s of kernels at 64 M rays):
allocation of the plane buffers, the ray-MLP chain, the sample gathers, the sample's chain, the norm pass -- each bracketed by
synchronize.  python tools/time_setup.py [gaussians] [rays_per_ellipsoid]"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SIXDGS_RANDOM_BACKBONE", "1")
pkg = importlib.import_module("6dgs_amd")
syn = importlib.import_module("6dgs_amd.synthetic")
ops = importlib.import_module("6dgs_amd.ops")
_lib = importlib.import_module("6dgs_amd._lib")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = "cuda"
scene = pkg.GaussianScene.from_dict(syn.make_scene(n, 0), device=dev)
ori, dr, rgb = pkg.generate_all_possible_rays(scene, max_ellipsoids=-1, emitter="isocell", rays_per_ellipsoid=k)
idm = pkg.IdentificationModule("dino")
idm.load_state_dict({kk: torch.from_numpy(v) for kk, v in syn.make_scorer_state_dict(0, with_cnn=True).items()}, strict=False)
idm = idm.to(dev).eval()
w = idm.packed_weights(ori.device)
r = ori.shape[0]


def t(label, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    print(f"{label:46s} {1e3 * (time.perf_counter() - t0):9.1f} ms", flush=True)
    return out


print(f"R = {r} rays")
t("warm: ray_keys of 300k rays (module load, packing)", lambda: ops.ray_keys(ori[:300000], dr[:300000], rgb[:300000], w, want_key=False, want_planes=True))
planes = t("torch.empty planes (1536 B/ray)", lambda: torch.empty(r, 1536, dtype=torch.uint8, device=dev))
del planes
nbytes = _lib.load().sixdgs_ray_keys_workspace_bytes(r, 262144)
ws = t(f"torch.empty ray_keys workspace ({nbytes / 1e9:.2f} GB)", lambda: torch.empty(nbytes, dtype=torch.uint8, device=dev))
del ws
res = t("ops.ray_keys main (alloc + chain)", lambda: ops.ray_keys(ori, dr, rgb, w, want_key=False, want_planes=True))
res2 = t("ops.ray_keys main again (allocator warm)", lambda: ops.ray_keys(ori, dr, rgb, w, want_key=False, want_planes=True))
del res2
si = t("select_sample_indices", lambda: ops.select_sample_indices(r, ori.device))
g = t("3 gathers rays[si]", lambda: (ori[si], dr[si], rgb[si]))
t("sample ray_keys", lambda: ops.ray_keys(g[0], g[1], g[2], w, want_key=False, want_planes=True))
t("key_norm_max", lambda: ops.key_norm_max(res[2][0], res[2][1]))
del res, g
torch.cuda.empty_cache()
t("idm._ensure_keys (cold allocator)", lambda: idm._ensure_keys(ori, dr, rgb))
idm.invalidate_caches()
idm._key_cache = None
t("idm._ensure_keys (again, allocator warm)", lambda: idm._ensure_keys(ori, dr, rgb))
