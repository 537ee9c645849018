"""100 MHz stamps of k_solve_pose's sections for image 0 (developer build: python tools/build_variant.py poseprof -DSDG_POSE_PROF; SIXDGS_LIB=...)."""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
ops = importlib.import_module("6dgs_amd.ops")
lib = importlib.import_module("6dgs_amd._lib").load()
lib.sixdgs_debug_pose_prof.argtypes = [C.c_void_p]
torch.manual_seed(0)
R, B, K = 1_000_000, 4, 100
ori = torch.randn(R, 3, device="cuda"); d = torch.nn.functional.normalize(torch.randn(R, 3, device="cuda"), dim=1)
idx = torch.stack([torch.randperm(R, device="cuda")[:K] for _ in range(B)]); w = torch.rand(B, K, device="cuda")
up = torch.nn.functional.normalize(torch.randn(B, 3, device="cuda"), dim=1)
for _ in range(3):
    ops.solve_pose(ori, d, idx, w, up, None)
torch.cuda.synchronize()
buf = (C.c_longlong * 16)()
lib.sixdgs_debug_pose_prof(buf)
s = list(buf)
names = ["gather", "duplicate counts + unique list", "isin", "compaction + per-ray terms", "weight sum, R, q sums, solve", "exclusion, renormalise, direction sum", "rotation, outputs"]
print(" | ".join(f"{n} {(s[i + 1] - s[i]) / 100:.2f} us" for i, n in enumerate(names)), f"| whole {(s[7] - s[0]) / 100:.2f} us")
