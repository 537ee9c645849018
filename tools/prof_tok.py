"""Cycle stamps of k_tok_gemm's sections (developer build: python tools/build_variant.py tokprof -DSDG_TOK_PROF; SIXDGS_LIB=build/variants/lib_tokprof.so):
workgroup (0, 0), thread 0, constant 100 MHz counter.  Sections: weight prefetch issued + LayerNorm parameters | staging of the token tile (last chunk) |
slab loop (last chunk) | epilogue (last feature tile).  python tools/prof_tok.py"""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

ops = importlib.import_module("6dgs_amd.ops")
lib = importlib.import_module("6dgs_amd._lib").load()
lib.sixdgs_debug_tok_prof.argtypes = [C.c_void_p]


def stamps():
    buf = (C.c_longlong * 16)()
    torch.cuda.synchronize()
    lib.sixdgs_debug_tok_prof(buf)
    return list(buf)


def main():
    dev = "cuda"
    torch.manual_seed(0)
    for images in (1, 16):
        m = 257 * images
        x = torch.randn(m, 384, device=dev)
        hid = torch.randn(m, 1536, device=dev)
        lw, lb = torch.rand(384, device=dev) + 0.5, torch.randn(384, device=dev) * 0.1
        w1, b1 = torch.randn(1536, 384, device=dev) * 0.05, torch.randn(1536, device=dev)
        wp, bp = torch.randn(384, 384, device=dev) * 0.05, torch.randn(384, device=dev)
        w2, b2 = torch.randn(384, 1536, device=dev) * 0.05, torch.randn(384, device=dev)
        cases = [("LN+FC1+GELU", lambda: ops.tok_linear(x, w1, b1, ln=(lw, lb, 1e-6), epilogue=ops.TOK_EPI_GELU)),
                 ("plain 384->384 + resid", lambda: ops.tok_linear(x, wp, bp, epilogue=ops.TOK_EPI_RESID, residual=x)),
                 ("FC2 + resid", lambda: ops.tok_linear(hid, w2, b2, epilogue=ops.TOK_EPI_RESID, residual=x))]
        for name, fn in cases:
            for _ in range(3):
                fn()
            s = stamps()
            d = lambda a, b: (s[b] - s[a]) / 100.0
            print(f"images {images:2d} {name:24s}: start->stage {d(0, 1):6.2f} us | stage {d(1, 2):6.2f} | slab loop {d(2, 3):6.2f} | epilogue {d(3, 4):6.2f} | whole workgroup {d(0, 4):6.2f}")


if __name__ == "__main__":
    main()
