"""Time the second scorer pass (k_score_reduce_blocked + top-k) alone on logits left by pass 1."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("6dgs_amd.ops")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 32_000_000
B = 2
torch.manual_seed(0)
planes = torch.empty(R, 1536, dtype=torch.uint8, device="cuda")
chunk = 2_000_000
inv = torch.empty((R + 127) // 128, device="cuda")
for r0 in range(0, R, chunk):
    k = torch.randn(min(chunk, R - r0), 384, device="cuda")
    p, s = ops.split_planes_f16(k)
    planes[r0:r0 + k.shape[0]] = p
    inv[r0 // 128:r0 // 128 + s.shape[0]] = s
q = torch.randn(B, 256, 384, device="cuda")
n_tok = torch.full((B,), 256, dtype=torch.int32, device="cuda")
ws = torch.empty(ops.score_topk_workspace_bytes(R, B, 100), dtype=torch.uint8, device="cuda")
st = ops.score_pass1(q, n_tok, None, ws, 100, key_planes=planes, key_scale=inv)
torch.cuda.synchronize()
for it in range(3):
    t0 = time.perf_counter()
    for _ in range(5):
        idx, val, sc = ops.score_pass2(st, n_tok, R, ws, 100, used_planes=True, want_scores=False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / 5
    bpr = 784 if ops.effective_mma_mode() == ops.MMA_F16X3 else 1024      # bytes of logits per ray and image (24-bit + references / fp32)
    print(f"pass 2 (reduce + top-k), {B} images x {R} rays: {ms:.2f} ms  -> logits stream {B * R * bpr / ms / 1e9:.2f} TB/s incl. top-k")
