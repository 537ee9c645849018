"""Is the 256 MB Infinity Cache visible to a write-then-read working set?  (a) copy bandwidth over footprints from 32 MB to 4 GB;
(b) the ray-MLP chain at chunk sizes whose workspace (4776 B per ray) does / does not fit it."""
import importlib, os, sys, time
os.environ.setdefault("SIXDGS_RANDOM_BACKBONE", "1")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
syn = importlib.import_module("6dgs_amd.synthetic"); ops = importlib.import_module("6dgs_amd.ops")

def ev(fn, n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 4096):
    n = mb * (1 << 20) // 4
    x = torch.empty(n // 2, device="cuda").normal_(); y = torch.empty_like(x)
    ms = ev(lambda: y.copy_(x), 20)
    ms2 = ev(lambda: x.mul_(1.0001), 20)
    print(f"footprint {mb:5d} MB: copy {mb / 1e3 / (ms * 1e-3) / 1e3:6.2f} TB/s   in-place (half the footprint, r+w) {mb / 1e3 / (ms2 * 1e-3) / 1e3:6.2f} TB/s", flush=True)

R = 1 << 21
rays = syn.make_rays(1 << 20, 0)
o, d, c = (torch.from_numpy(rays[k]).cuda().repeat(2, 1)[:R].contiguous() for k in ("ori", "dir", "rgb"))
w = ops.PackedWeights({k: torch.from_numpy(v) for k, v in syn.make_scorer_state_dict(0).items()}, "cuda")
for chunk in (16384, 32768, 49152, 65536, 131072, 262144, 524288):
    ms = ev(lambda: ops.ray_keys(o, d, c, w, want_key=False, want_planes=True, max_chunk=chunk), 3)
    print(f"chain, chunk {chunk:7d} rays (workspace {chunk * 4776 / 1e6:7.1f} MB): {ms:7.2f} ms for {R} rays = {R * 2025472 / ms / 1e9:6.1f} TFLOP/s", flush=True)
