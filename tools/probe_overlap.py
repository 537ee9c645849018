"""Does the HBM-bound second pass of image group g overlap with the power-bound logits pass of group g+1 on a second stream?"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("6dgs_amd.ops")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 32_000_000
B = 2
torch.manual_seed(0)
planes = torch.empty(R, 1536, dtype=torch.uint8, device="cuda")
inv = torch.empty((R + 127) // 128, device="cuda")
chunk = 2_000_000
for r0 in range(0, R, chunk):
    k = torch.randn(min(chunk, R - r0), 384, device="cuda")
    p, s = ops.split_planes_f16(k)
    planes[r0:r0 + k.shape[0]] = p
    inv[r0 // 128:r0 // 128 + s.shape[0]] = s
q = [torch.randn(B, 256, 384, device="cuda") for _ in range(2)]
n_tok = torch.full((B,), 256, dtype=torch.int32, device="cuda")
ws = [torch.empty(ops.score_topk_workspace_bytes(R, B, 100), dtype=torch.uint8, device="cuda") for _ in range(2)]
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

def p1(g):
    return ops.score_pass1(q[g], n_tok, None, ws[g], 100, key_planes=planes, key_scale=inv)

def p2(g, st):
    return ops.score_pass2(st, n_tok, R, ws[g], 100, used_planes=True, want_scores=False)

def sequential():
    with torch.cuda.stream(sa):
        st0 = p1(0); r0 = p2(0, st0); st1 = p1(1); r1 = p2(1, st1)
    return r0, r1

def overlapped():
    with torch.cuda.stream(sa):
        st0 = p1(0)
        e0 = torch.cuda.Event(); e0.record(sa)
        st1 = p1(1)
    with torch.cuda.stream(sb):
        sb.wait_event(e0)
        r0 = p2(0, st0)
        e1 = torch.cuda.Event(); e1.record(sb)
    with torch.cuda.stream(sa):
        r1 = p2(1, st1)
        sa.wait_event(e1)
    return r0, r1

for name, fn in (("sequential", sequential), ("overlapped", overlapped), ("sequential", sequential), ("overlapped", overlapped)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        a, b = fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) * 1e3 / 3:.2f} ms per 4 images")
ra, rb = sequential(); oa, ob = overlapped(); torch.cuda.synchronize()
print("same results:", torch.equal(ra[0], oa[0]) and torch.equal(rb[0], ob[0]) and torch.equal(ra[1], oa[1]))
