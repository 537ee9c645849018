"""Timing-only ablations of the DMA-fed logits kernel (SIXDGS_DEBUG_ABLATE): which part of a tile costs what."""
import importlib, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# private build of the library with the ablation switches compiled in (never the shipped .so)
abl_so = os.path.join(ROOT, "gpurun_out", "lib6dgs_hip_ablation.so")
if not os.path.exists(abl_so):
    os.makedirs(os.path.dirname(abl_so), exist_ok=True)
    csrc = os.path.join(ROOT, "6dgs_amd", "csrc")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-DSIXDGS_ABLATION",
                           "-shared", *[os.path.join(csrc, f) for f in ("geometry.hip", "gemm.hip", "score.hip", "pose.hip")], "-o", abl_so])
L = importlib.import_module("6dgs_amd._lib")
L.LIB_PATH = abl_so
ops = importlib.import_module("6dgs_amd.ops")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 6_400_000
torch.manual_seed(0)
key = torch.randn(R, 384, device="cuda")
F16 = True                      # (the bf16 x 6 plane scorer this tool also timed was removed in round 6)
ops.set_mma_mode(ops.MMA_F16X3)
planes, kscale = ops.split_planes_f16(key)
del key
q = torch.randn(2, 256, 384, device="cuda")
n_tok = torch.full((2,), 256, dtype=torch.int32, device="cuda")
ws = torch.empty(ops.score_topk_workspace_bytes(R, 2, 100), dtype=torch.uint8, device="cuda")
names = {2048: "cycle stamps", 512: "nt key DMA", 1024: "nt q DMA", 1536: "nt key+q DMA", 256: "plain (temporal) stores", 64: "no logit stores", 128: "no exp/sum stats", 192: "no stores, no exp/sum", 3: "no Q DMA, no epilogue", 10: "no key DMA, no epilogue", 59: "MFMA only (no DMA/reads/epilogue/barriers)", 0: "full", 1: "Q DMA first tile only", 8: "key DMA first tile only", 9: "no DMA after first tile", 2: "no epilogue",
         4: "no MFMA", 13: "no MFMA, no DMA after first tile", 6: "no MFMA, no epilogue", 11: "no DMA, no epilogue", 27: "MFMA + barriers only (no DMA/reads/epilogue)", 18: "no frag reads, no epilogue"}
ABLS = tuple(int(a) for a in os.environ.get("ABLATE_LIST", "0,2,9,11,64,27,59,2048").split(","))
for abl in ABLS:
    os.environ["SIXDGS_DEBUG_ABLATE"] = str(abl)
    for it in range(2):
        prof = ops.KernelProfile()
        ops.score_topk(q, n_tok, None, 100, want_scores=False, workspace=ws, key_planes=planes, key_scale=kscale, profile=prof, n_tok_host=[256, 256])
        ms, fl, by, n = prof.collect()
    print(f"ABL={abl:3d} {names[abl]:34s} {ms:8.2f} ms   {fl / ms / 1e9:7.1f} TFLOP/s-eq   per WG-tile {ms * 1e3 * 256 / (R / 128 * 4):6.2f} us")


if 2048 not in ABLS:
    sys.exit(0)
import ctypes, numpy as np
lib = ctypes.CDLL(abl_so)
buf = (ctypes.c_ulonglong * 64)()
torch.cuda.synchronize()
lib.sixdgs_debug_cycles(buf, 1)
os.environ["SIXDGS_DEBUG_ABLATE"] = "2048"
ops.score_topk(q, n_tok, None, 100, want_scores=False, workspace=ws, key_planes=planes, key_scale=kscale, n_tok_host=[256, 256])
torch.cuda.synchronize()
lib.sixdgs_debug_cycles(buf, 0)
a = np.array(list(buf), dtype=np.float64).reshape(8, 8)
nblk = a[0, 7]
print("blocks sampled", nblk)
ph = ["loop", "steps 0-2", "vmcnt wait", "barrier", "step 3", "epilogue"]
tot = a[:, :6].sum(axis=1)
for w in range(8):
    print(f"wave {w}: total {tot[w] / nblk:12.0f} cyc/block  " + "  ".join(f"{ph[i]} {100 * a[w, i] / tot[w]:5.1f}%" for i in range(6)))
