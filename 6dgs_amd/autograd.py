"""Differentiable dense layers of the scorer on the hand-written MFMA GEMM (backward for the ray MLP and
q_proj / k_proj), for the on-box training of IdentificationModule (pose_estimation/train.py:106-170 drives
identification_module.py:94-115 -> ray_preprocessor.py:36-46 and our_multihead_attention.py:70-79 through autograd).

One torch.autograd.Function, `HipLinear`: y = act(x w^T + b) with all three products of the layer on `sixdgs_linear`
(gemm.hip: 128x128 tiles, fp32 operands as 3 bf16 planes x 6 MFMA terms, fp32-equivalent):

    forward   y   = x  . w^T            [M,K] x [N,K] -> [M,N]     (bias + ReLU in the kernel's epilogue)
    backward  dx  = dy . w              as dy [M,N] x (w^T) [K,N]   -> [M,K]   (when K is a multiple of 128: the kernel's N constraint)
              dw^T = x^T . dy           as x^T [K,M] x (dy^T) [N,M] -> [K,N]   (M zero-padded to a multiple of 16: the kernel's K constraint)
              db  = column sums of dy   (PyTorch reduction)

The kernel wants K % 16 == 0 and N % 128 == 0, so the callers split / pad the reference's odd widths (141, 653, 398) with zeros --
exact in value.  dx for a K that is not a multiple of 128 is only ever needed for inputs without gradient on this path (ray
encodings, image tokens of a frozen backbone); if it is requested anyway it is formed by a PyTorch matmul, said so in the name.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import ops


class HipLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, relu: bool):
        y = ops.linear(x, w, b, relu=relu, split_k=1)
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.relu, ctx.has_bias = relu, b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = dy.contiguous()
        if ctx.relu:
            dy = dy * (y > 0)
        dx = dw = db = None
        m, k = x.shape
        if ctx.needs_input_grad[0]:
            dx = ops.linear(dy, w.t().contiguous(), None, split_k=1) if k % 128 == 0 else dy @ w      # second form: PyTorch matmul (rocBLAS)
        if ctx.needs_input_grad[1]:
            pad = (-m) % 16
            xt, dyt = x.t().contiguous(), dy.t().contiguous()
            if pad:
                xt, dyt = F.pad(xt, (0, pad)), F.pad(dyt, (0, pad))
            dw = ops.linear(xt, dyt, None, split_k=1).t()                # [K,N] -> [N,K]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(dim=0)
        return dx, dw, db, None


def linear(x: torch.Tensor, weight: torch.Tensor, bias, relu: bool = False) -> torch.Tensor:
    """act(x weight^T + bias) through HipLinear; the input width is zero-padded to a multiple of 16 when needed."""
    k = x.shape[-1]
    pad = (-k) % 16
    if pad:
        x, weight = F.pad(x, (0, pad)), F.pad(weight, (0, pad))
    return HipLinear.apply(x.contiguous(), weight.contiguous(), bias, relu)


def ray_mlp(rp, x: torch.Tensor) -> torch.Tensor:
    """RayPreprocessor.forward (ray_preprocessor.py:36-46) on the encoded input x [R,141]: mlp, then mlp2 on cat([h, x]) -- evaluated as
    h W3[:, :512]^T + x W3[:, 512:]^T so that the gradient with respect to h (K = 512) runs on the kernel too."""
    h = linear(x, rp.mlp[0].weight, rp.mlp[0].bias, relu=True)
    h = linear(h, rp.mlp[2].weight, rp.mlp[2].bias, relu=True)
    w3, hid = rp.mlp2[0].weight, rp.mlp[2].weight.shape[0]
    z = linear(h, w3[:, :hid], rp.mlp2[0].bias) + linear(x, w3[:, hid:], None)
    return linear(torch.relu(z), rp.mlp2[2].weight, rp.mlp2[2].bias)
