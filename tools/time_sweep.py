"""Times the select path's main sweep alone (sixdgs_select_sweep: k_logits_f16x<kOutUB>) for B images x R rays of random key planes:
    python tools/time_sweep.py <B> <R> [reps] [--tokens T]     (SIXDGS_SIBLING_SYNC = 0 / 1 / 2 selects the grid layout; T tokens per image, default 256:
    masked views keep 56-140 -- the kernel's waves beyond the token count skip their MFMAs)"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("6dgs_amd.ops")
T = 256
if "--tokens" in sys.argv:
    i = sys.argv.index("--tokens")
    T = int(sys.argv[i + 1])
    del sys.argv[i:i + 2]
B, R = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
torch.manual_seed(0)
planes = torch.empty(R, 1536, dtype=torch.uint8, device="cuda")
chunk = 1 << 20
for r0 in range(0, R, chunk):                     # random keys, split chunk by chunk (the fp32 keys of 32 M rays would be 49 GB more)
    k = torch.randn(min(chunk, R - r0), 384, device="cuda") * 0.07
    p, s = ops.split_planes_f16(k)
    planes[r0:r0 + k.shape[0]] = p
    if r0 == 0:
        scale = torch.empty((R + 127) // 128, device="cuda")
    scale[r0 // 128:r0 // 128 + s.shape[0]] = s
q = torch.randn(B, 256, 384, device="cuda")
q[:, T:] = 0
nt = torch.full((B,), T, dtype=torch.int32, device="cuda")
si = ops.select_sample_indices(min(R, 1 << 22), "cuda")
sp, ssc = planes[si].contiguous(), None
_, ssc = ops.split_planes_f16(torch.randn(si.shape[0], 384, device="cuda") * 0.07)
ss = ops.SelectStream(q, nt, R, 100, 4096, [T] * B)
ss.begin(sp, ssc)
ts = []
for it in range(reps + 1):
    prof = ops.KernelProfile()
    ss.sweep(planes, scale, 0, prof, update_norm=False)
    ms, fl, by, n = prof.collect()
    if it:
        ts.append(ms)
ts.sort()
med = ts[len(ts) // 2]
print(f"sweep B={B} R={R} tokens={T} mode={os.environ.get('SIXDGS_SIBLING_SYNC', 'default')}: median {med:.3f} ms (min {ts[0]:.3f}, max {ts[-1]:.3f}) = {fl / med / 1e9:.1f} TFLOP/s fp32-eq, frac {fl / med / 1e9 / (2500 / 3):.4f}")
