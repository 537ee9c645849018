"""Times the ray MLP + k_proj chain (sixdgs_ray_keys_ex -> key planes) on R synthetic rays: ms and fp32-equivalent TFLOP/s (2 025 472 FLOP per ray)."""
import importlib, os, sys
os.environ.setdefault("SIXDGS_RANDOM_BACKBONE", "1")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
syn = importlib.import_module("6dgs_amd.synthetic"); ops = importlib.import_module("6dgs_amd.ops")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4_194_304
CHUNK = int(sys.argv[2]) if len(sys.argv) > 2 else ops.RAY_KEYS_CHUNK
rays = syn.make_rays(min(R, 1 << 20), 0)
o, d, c = (torch.from_numpy(rays[k]).cuda().repeat((R + (1 << 20) - 1) // (1 << 20), 1)[:R].contiguous() for k in ("ori", "dir", "rgb"))
w = ops.PackedWeights({k: torch.from_numpy(v) for k, v in syn.make_scorer_state_dict(0).items()}, "cuda")
for name, kw in (("planes only (plane-to-plane chain)", dict(want_key=False, want_planes=True)), ("features + keys (fp32-operand kernels)", dict(want_feat=True, want_key=True))):
    ts = []
    for it in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out = ops.ray_keys(o, d, c, w, max_chunk=CHUNK, **kw); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b)); del out
    ms = min(ts[1:])
    print(f"ray_keys R={R} chunk={CHUNK} {name}: {ms:.2f} ms = {R * 2025472 / ms / 1e9:.1f} TFLOP/s fp32-equivalent")
