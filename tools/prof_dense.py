"""Developer tool: cycle stamps of k_dense_planes' loop sections (needs a library built with SIXDGS_EXTRA_FLAGS=-DSDG_DENSE_PROF).
Runs the plane-to-plane chain on one chunk of rays and prints, per wave of one workgroup of the LAST layer launched, the cycles spent
issuing MFMAs (incl. LDS fragment waits), staging (global-load waits + LDS writes), issuing fetches, in the barrier and in the epilogue."""
import ctypes, importlib, os, sys
os.environ.setdefault("SIXDGS_RANDOM_BACKBONE", "1")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
syn = importlib.import_module("6dgs_amd.synthetic"); ops = importlib.import_module("6dgs_amd.ops"); lib = importlib.import_module("6dgs_amd._lib")
R = 262144
rays = syn.make_rays(R, 0)
o, d, c = (torch.from_numpy(rays[k]).cuda() for k in ("ori", "dir", "rgb"))
w = ops.PackedWeights({k: torch.from_numpy(v) for k, v in syn.make_scorer_state_dict(0).items()}, "cuda")
for _ in range(2):
    ops.ray_keys(o, d, c, w, want_key=False, want_planes=True)
torch.cuda.synchronize()
h = ctypes.CDLL(lib.LIB_PATH)
buf = (ctypes.c_longlong * 128)()
assert h.sixdgs_debug_dense_prof(buf) == 0
print("wave   slab loop   epilogues | total cycles, wall us, GHz   (one persistent workgroup of the last layer: k_proj, 12 slabs, one pass of 384 features, 8 tiles of 128 rays)")
for wv in range(8):
    v = buf[wv * 16: wv * 16 + 16]
    print(f"{wv:4d} {v[0]:11d} {v[1]:11d} | {v[8]:9d} {v[9] / 100:8.1f} {v[8] / max(v[9], 1) / 10:5.2f}")
