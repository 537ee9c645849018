"""Round 6: the four dense stages of a ViT-S/14 block -- PyTorch-ROCm's kernels (hipBLASLt fp32 GEMM + the elementwise kernels around
it) against this build's sixdgs_tok_linear (packed fp16 weight planes, LayerNorm / GELU / residual / head layouts folded in) -- at the token counts of
1 / 4 / 16 / 64 images (M = 257 x images).  Every figure is GPU time per stage INSIDE a hipGraph of 24 back-to-back repetitions (the host's launch rate
-- ~19 us per PyTorch call -- does not enter), then the whole forward as a graph.  python tools/time_vit_gemms.py"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

ops = importlib.import_module("6dgs_amd.ops")
REPS = 24


def graph_us(fn, replays=20):
    """GPU microseconds per call of fn, measured over a captured graph of REPS calls."""
    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s_)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REPS):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(replays):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (replays * REPS) * 1e3


def main():
    torch.manual_seed(0)
    dev = "cuda"
    print("| images | M | stage | PyTorch kernels us | sixdgs_tok_linear us | own / PyTorch | TFLOP/s PyTorch | TFLOP/s own | max err own vs fp64 |")
    print("|---:|---:|---|---:|---:|---:|---:|---:|---:|")
    for images in (1, 2, 4, 8, 16, 64):
        m, heads, tokens = 257 * images, 6, 257
        x = torch.randn(m, 384, device=dev)
        lw, lb = torch.rand(384, device=dev) + 0.5, torch.randn(384, device=dev) * 0.1
        gam = torch.rand(384, device=dev) * 0.5 + 0.1
        att = torch.randn(m, 384, device=dev)
        hid = torch.randn(m, 1536, device=dev)
        W = lambda n, k: torch.randn(n, k, device=dev) * 0.05
        wq, bq, wp, bp, w1, b1, w2, b2 = W(1152, 384), torch.randn(1152, device=dev), W(384, 384), torch.randn(384, device=dev), W(1536, 384), torch.randn(1536, device=dev), W(384, 1536), torch.randn(384, device=dev)
        stages = [
            ("LayerNorm + QKV (384 -> 1152)", 2.0 * m * 384 * 1152,
             lambda: F.linear(F.layer_norm(x, (384,), lw, lb, 1e-6), wq, bq),
             lambda: ops.tok_linear(x, wq, bq, ln=(lw, lb, 1e-6)),
             lambda: F.linear(F.layer_norm(x.double(), (384,), lw.double(), lb.double(), 1e-6), wq.double(), bq.double())),
            ("proj (384 -> 384) + LayerScale + residual", 2.0 * m * 384 * 384,
             lambda: torch.addcmul(x, F.linear(att, wp, bp), gam),
             lambda: ops.tok_linear(att, wp, bp, epilogue=ops.TOK_EPI_RESID, residual=x, gamma=gam),
             lambda: x.double() + gam.double() * F.linear(att.double(), wp.double(), bp.double())),
            ("LayerNorm + FC1 (384 -> 1536) + GELU", 2.0 * m * 384 * 1536,
             lambda: F.gelu(F.linear(F.layer_norm(x, (384,), lw, lb, 1e-6), w1, b1)),
             lambda: ops.tok_linear(x, w1, b1, ln=(lw, lb, 1e-6), epilogue=ops.TOK_EPI_GELU),
             lambda: F.gelu(F.linear(F.layer_norm(x.double(), (384,), lw.double(), lb.double(), 1e-6), w1.double(), b1.double()))),
            ("FC2 (1536 -> 384) + LayerScale + residual", 2.0 * m * 1536 * 384,
             lambda: torch.addcmul(x, F.linear(hid, w2, b2), gam),
             lambda: ops.tok_linear(hid, w2, b2, epilogue=ops.TOK_EPI_RESID, residual=x, gamma=gam),
             lambda: x.double() + gam.double() * F.linear(hid.double(), w2.double(), b2.double())),
        ]
        tl = to = 0.0
        for name, fl, lib, own, ref in stages:
            a, b = graph_us(lib), graph_us(own)
            r = ref()
            err = float((own().double() - r).abs().max() / r.abs().max())
            tl, to = tl + a, to + b
            print(f"| {images} | {m} | {name} | {a:.1f} | {b:.1f} | {b / a:.2f} | {fl / a / 1e6:.1f} | {fl / b / 1e6:.1f} | {err:.1e} |")
        qkv = torch.randn(m, 1152, device=dev)
        q_, k_, v_ = qkv.view(images, tokens, 3, heads, 64).permute(2, 0, 3, 1, 4).unbind(0)
        at = graph_us(lambda: F.scaled_dot_product_attention(q_, k_, v_).transpose(1, 2).reshape(m, 384))
        ao = graph_us(lambda: ops.tok_attention(qkv, images, tokens, heads))
        qd, kd, vd = qkv.double().view(images, tokens, 3, heads, 64).permute(2, 0, 3, 1, 4).unbind(0)
        r = F.scaled_dot_product_attention(qd, kd, vd).transpose(1, 2).reshape(m, 384)
        aerr = float((ops.tok_attention(qkv, images, tokens, heads).double() - r).abs().max() / r.abs().max())
        print(f"| {images} | {m} | the four dense stages of a block | {tl:.1f} | {to:.1f} | {to / tl:.2f} | | | |")
        print(f"| {images} | {m} | attention (F.scaled_dot_product_attention on strided views vs sixdgs_tok_attention) | {at:.1f} | {ao:.1f} | {ao / at:.2f} | | | {aerr:.1e} |")
    # the whole ViT-S/14 forward as a hipGraph, blocks fused (five launches) and not
    bb = importlib.import_module("6dgs_amd.backbone")
    vit = bb.ViTS14().eval().cuda()
    print()
    print("| images | ViT-S/14 forward as a hipGraph: PyTorch kernels us | every stage through sixdgs_tok_linear us | stages chosen by row count (default) us | default / PyTorch |")
    print("|---:|---:|---:|---:|---:|")
    global REPS
    REPS = 1
    for images in (1, 2, 4, 8, 16):
        x = torch.randn(images, 3, 224, 224, device=dev)
        res = {}
        for fused in ("0", "all", "1"):
            os.environ["SIXDGS_VIT_FUSED"] = fused
            with torch.no_grad():
                res[fused] = graph_us(lambda: vit.forward_features(x), 40)
        print(f"| {images} | {res['0']:.0f} | {res['all']:.0f} | {res['1']:.0f} | {res['1'] / res['0']:.2f} |")


if __name__ == "__main__":
    main()
