"""Do two scorer calls on two HIP streams overlap (logits kernel of one with the HBM-bound reduce of the other)?"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("6dgs_amd.ops")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 6_400_000
torch.manual_seed(0)
key = torch.randn(R, 384, device="cuda")
planes, kscale = ops.split_planes_f16(key)
del key
q = torch.randn(2, 256, 384, device="cuda")
n_tok = torch.full((2,), 256, dtype=torch.int32, device="cuda")
ws = [torch.empty(ops.score_topk_workspace_bytes(R, 2, 100), dtype=torch.uint8, device="cuda") for _ in range(2)]
st = [torch.cuda.Stream() for _ in range(2)]

def run(n_calls, two_streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_calls):
        with torch.cuda.stream(st[i % 2] if two_streams else st[0]):
            ops.score_topk(q, n_tok, None, 100, want_scores=False, workspace=ws[i % 2], key_planes=planes, key_scale=kscale, n_tok_host=[256, 256])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n_calls

for rep in range(2):
    print(f"one stream : {run(8, False):7.2f} ms per call     two streams: {run(8, True):7.2f} ms per call")
