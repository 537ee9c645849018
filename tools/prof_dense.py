"""Developer tool: cycle stamps of k_dense_planes' loop sections (needs a library built with SIXDGS_EXTRA_FLAGS=-DSDG_DENSE_PROF).
Runs the plane-to-plane chain on one chunk of rays and prints, per wave of one workgroup of the LAST layer launched, the cycles spent
issuing MFMAs (incl. LDS fragment waits), staging (global-load waits + LDS writes), issuing fetches, in the barrier and in the epilogue."""
import ctypes, importlib, os, sys
os.environ.setdefault("SIXDGS_RANDOM_BACKBONE", "1")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
syn = importlib.import_module("6dgs_amd.synthetic"); ops = importlib.import_module("6dgs_amd.ops"); lib = importlib.import_module("6dgs_amd._lib")
R = 1048576
rays = syn.make_rays(R, 0)
o, d, c = (torch.from_numpy(rays[k]).cuda() for k in ("ori", "dir", "rgb"))
w = ops.PackedWeights({k: torch.from_numpy(v) for k, v in syn.make_scorer_state_dict(0).items()}, "cuda")
h = ctypes.CDLL(lib.LIB_PATH)
buf = (ctypes.c_longlong * 128)()
# one persistent workgroup (the grid's middle one), per wave: cycles in the flattened slab loop incl. barriers, cycles in the epilogues, total; the chain
# is stopped behind layer n (SIXDGS_DENSE_PROF_LAYER, honoured by the profiling build only) so that the stamps are that layer's
for layer, what in ((1, "layer 1: 5 slabs, 256 x 256 tile"), (2, "layer 2: 16 slabs"), (3, "layer 3: 21 slabs"), (0, "layer 4 + k_proj: 16 slabs, 384 x 128 tile, one pass")):
    os.environ["SIXDGS_DENSE_PROF_LAYER"] = str(layer)
    for _ in range(2):
        ops.ray_keys(o, d, c, w, want_key=(layer != 0), want_planes=(layer == 0)) if False else ops.ray_keys(o, d, c, w, want_key=False, want_planes=True)
    torch.cuda.synchronize()
    assert h.sixdgs_debug_dense_prof(buf) == 0
    passes = (R // 256) * 2 // 256 if layer else (R // 128) // 256
    print(f"{what}; {passes} (tile, pass) items per workgroup at R = {R}")
    print("wave   slab loop   epilogues | total cycles, wall us, GHz | per item: loop us, epilogue us")
    for wv in range(8):
        v = buf[wv * 16: wv * 16 + 16]
        ghz = v[8] / max(v[9], 1) / 10
        print(f"{wv:4d} {v[0]:11d} {v[1]:11d} | {v[8]:9d} {v[9] / 100:8.1f} {ghz:5.2f} | {v[0] / ghz / 1e3 / max(passes, 1):6.2f} {v[1] / ghz / 1e3 / max(passes, 1):6.2f}")
