"""BackboneWrapper -- mirror of pose_estimation/backbone.py:34-139: resize 256 (bicubic, antialias)
-> centre crop 224 -> ImageNet normalisation -> DINOv2 ViT-S/14 patch tokens [256,384] + 14-channel
grid position encoding -> tokens [T<=256, 398], selected by the alpha mask.

Stays on PyTorch-ROCm (frozen third-party ViT): its output is the Q-side INPUT of
the HIP boundary.  torchvision is not required: its tensor Resize/CenterCrop are F.interpolate
(antialias) and a slice, reproduced here.

DINOv2 weights come from torch.hub in the reference (backbone.py:15, network access).  Offline the
factory falls back to a randomly initialised ViT-S/14 of the same architecture (module names follow
the dinov2 state_dict so real weights load when available); pass `backbone=` to inject your own
module exposing forward_features(x)["x_norm_patchtokens"].
"""
from __future__ import annotations

import math
import os
import warnings
from typing import Optional

import torch
import torch.nn.functional as F

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


# ---- minimal ViT-S/14 with dinov2's parameter names ------------------------------------------------
class _Attention(torch.nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = torch.nn.Linear(dim, dim * 3)
        self.proj = torch.nn.Linear(dim, dim)

    def forward(self, x):
        b, n, c = x.shape
        qkv = self.qkv(x).reshape(b, n, 3, self.num_heads, c // self.num_heads).permute(2, 0, 3, 1, 4)
        y = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])
        return self.proj(y.transpose(1, 2).reshape(b, n, c))


class _LayerScale(torch.nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.gamma = torch.nn.Parameter(torch.ones(dim))

    def forward(self, x):
        return x * self.gamma


class _Mlp(torch.nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = torch.nn.Linear(dim, hidden)
        self.fc2 = torch.nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class _Block(torch.nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.norm1 = torch.nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim, heads)
        self.ls1 = _LayerScale(dim)
        self.norm2 = torch.nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, dim * 4)
        self.ls2 = _LayerScale(dim)

    def forward(self, x):
        if x.is_cuda and not torch.is_grad_enabled():
            # inference on the GPU: x + gamma * branch as ONE kernel (addcmul) instead of a multiply and an add -- the image side is ~7 us per launch
            # whatever the launch does, and these are 24 of its ~170 launches (same value to the last ulp: one rounding instead of two)
            x = torch.addcmul(x, self.attn(self.norm1(x)), self.ls1.gamma)
            return torch.addcmul(x, self.mlp(self.norm2(x)), self.ls2.gamma)
        x = x + self.ls1(self.attn(self.norm1(x)))
        return x + self.ls2(self.mlp(self.norm2(x)))


class _PatchConv(torch.nn.Conv2d):
    """Conv2d with kernel == stride (non-overlapping patches) evaluated as one GEMM over the unfolded patches.
    MIOpen has no tuned fp32 solver for the 14x14 / stride-14 patch embedding and falls back to its naive kernel
    (2.3 ms per call on MI355X, more than the rest of the ViT); same parameters and state_dict keys as the Conv2d."""

    def forward(self, x):
        p, q = self.kernel_size
        if self.stride != (p, q) or self.padding != (0, 0) or self.dilation != (1, 1) or self.groups != 1 or x.shape[2] % p or x.shape[3] % q:
            return super().forward(x)
        b, c, h, w = x.shape
        cols = x.reshape(b, c, h // p, p, w // q, q).permute(0, 2, 4, 1, 3, 5).reshape(b, (h // p) * (w // q), c * p * q)
        y = F.linear(cols, self.weight.reshape(self.out_channels, -1), self.bias)
        return y.transpose(1, 2).reshape(b, self.out_channels, h // p, w // q)


class _PatchEmbed(torch.nn.Module):
    def __init__(self, dim, patch):
        super().__init__()
        self.proj = _PatchConv(3, dim, kernel_size=patch, stride=patch)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class ViTS14(torch.nn.Module):
    """DINOv2 ViT-S/14 forward (no register tokens), 224x224 -> 256 patch tokens of width 384."""

    def __init__(self, dim=384, depth=12, heads=6, patch=14, pos_grid=37):
        super().__init__()
        self.patch_size = patch
        self.patch_embed = _PatchEmbed(dim, patch)
        self.cls_token = torch.nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = torch.nn.Parameter(torch.zeros(1, 1 + pos_grid * pos_grid, dim))
        self.mask_token = torch.nn.Parameter(torch.zeros(1, dim))
        self.blocks = torch.nn.ModuleList([_Block(dim, heads) for _ in range(depth)])
        self.norm = torch.nn.LayerNorm(dim, eps=1e-6)
        g = torch.Generator().manual_seed(1234)
        with torch.no_grad():
            self.pos_embed.copy_(0.02 * torch.randn(self.pos_embed.shape, generator=g))
            self.cls_token.copy_(1e-6 * torch.randn(self.cls_token.shape, generator=g))

    def _pos(self, n_side):
        """Position embedding for an n_side x n_side patch grid (dinov2's interpolate_pos_encoding: bicubic resize of the 37 x 37 table).
        The result depends on the parameter alone, not on the image: it is built once per (grid size, parameter version, device) --
        recomputing it cost 0.87 ms per forward on MI355X (PyTorch's bicubic kernel on a [1,384,37,37] tensor), more than half of the
        rest of the ViT at batch 1."""
        pe = self.pos_embed
        m = int(math.isqrt(pe.shape[1] - 1))
        if m == n_side:
            return pe
        # one entry per grid size (a dict): a captured hipGraph has the address of the tensor it was captured with baked in, so an
        # entry must stay alive while any other grid size is in use; an entry is replaced only when the parameter itself changed
        ver = (pe._version, pe.data_ptr(), str(pe.device), pe.dtype)
        cache = self.__dict__.setdefault("_pos_cache", {})
        hit = cache.get(n_side)
        if hit is not None and hit[0] == ver and not (torch.is_grad_enabled() and pe.requires_grad):
            return hit[1]
        patch = pe[:, 1:].reshape(1, m, m, -1).permute(0, 3, 1, 2)
        patch = F.interpolate(patch, size=(n_side, n_side), mode="bicubic", align_corners=False)
        out = torch.cat([pe[:, :1], patch.permute(0, 2, 3, 1).reshape(1, n_side * n_side, -1)], dim=1)
        if not (torch.is_grad_enabled() and pe.requires_grad):
            cache[n_side] = (ver, out.detach())
        return out

    def forward_features(self, x):
        b = x.shape[0]
        t = self.patch_embed(x)
        t = torch.cat([self.cls_token.expand(b, -1, -1), t], dim=1) + self._pos(x.shape[-1] // self.patch_size)
        if fused_blocks_usable(self, t):
            t = fused_blocks(self.blocks, t, owned=True)          # (t is this function's own temporary: the blocks may update it in place)
        else:
            for blk in self.blocks:
                t = blk(t)
        t = self.norm(t)
        return {"x_norm_clstoken": t[:, 0], "x_norm_patchtokens": t[:, 1:]}


# ---- the transformer blocks at inference on the GPU: five launches per block (round 6) -------------------------------------------------------------
# From how many token rows (257 x images) each dense stage of a block runs through sixdgs_tok_linear instead of PyTorch's kernels: GPU time per stage
# inside a hipGraph on MI355X, profiles/r06_vit_stages.md.  Below ~8 images the library's launch-bound 8 us GEMMs win on the two N = 384 products
# (proj, FC2: a 64-token tile gives them 10 .. 30 workgroups), while the two LayerNorm-fused ones win from 2 images (LN + FC1 + GELU always).
FUSED_MIN_ROWS = {"qkv": 2 * 257, "attn": 0, "proj": 8 * 257, "fc1": 0, "fc2": 16 * 257}


def fused_blocks_usable(vit, t) -> bool:
    """The fused form serves fp32 inference on the GPU of blocks with the ViT-S layout (width 384 = 6 heads of 64, dinov2's parameter names);
    SIXDGS_VIT_FUSED=0 keeps PyTorch's kernels (the comparison the tests make), =all sends every stage through sixdgs_tok_linear whatever the batch."""
    if not t.is_cuda or torch.is_grad_enabled() or t.dtype != torch.float32 or os.environ.get("SIXDGS_VIT_FUSED", "1") == "0":
        return False
    try:
        blk = vit.blocks[0]
        return (t.shape[-1] == 384 and blk.attn.num_heads * 64 == 384 and tuple(blk.attn.qkv.weight.shape) == (1152, 384)
                and blk.mlp.fc1.weight.shape[1] == 384 and blk.mlp.fc2.weight.shape[0] == 384 and blk.attn.qkv.bias is not None)
    except AttributeError:
        return False


def fused_blocks(blocks, t: torch.Tensor, owned: bool = False) -> torch.Tensor:
    """x + ls1(attn(norm1(x))), then x + ls2(mlp(norm2(x))) for every block (_Block.forward; dinov2's NestedTensorBlock at inference) as FIVE launches:
         LayerNorm + QKV product + bias                              (sixdgs_tok_linear: A_LAYERNORM, EPI_BIAS)
         attention on that [M, 1152] matrix as it lies, token-major out  (sixdgs_tok_attention; F.scaled_dot_product_attention on strided views otherwise)
         proj + bias, LayerScale, residual                           (A_PLAIN, EPI_RESID, in place on the residual stream)
         LayerNorm + FC1 + bias + GELU                               (A_LAYERNORM, EPI_GELU)
         FC2 + bias, LayerScale, residual                            (A_PLAIN, EPI_RESID, in place)
    instead of twelve (2 LayerNorm, 4 GEMM, attention, GELU, 2 addcmul, 2 layout copies); a stage whose token matrix is below its row count in
    FUSED_MIN_ROWS keeps PyTorch's kernels.  The weights are split into fp16 planes once (ops.TokWeights, re-packed when a parameter changes); fp32-class
    results (two planes x three terms, fp32 accumulation)."""
    from . import ops
    b, n, c = t.shape
    m = b * n
    force = os.environ.get("SIXDGS_VIT_FUSED", "1") == "all"
    own = {k: force or m >= v for k, v in FUSED_MIN_ROWS.items()}
    x = t.reshape(m, c).contiguous()
    if x.data_ptr() == t.data_ptr() and not owned:
        x = x.clone()          # the residual stream is updated in place: never the caller's tensor
    one = None
    for blk in blocks:
        h = blk.attn.num_heads
        g1 = blk.ls1.gamma if hasattr(blk, "ls1") and hasattr(blk.ls1, "gamma") else None
        g2 = blk.ls2.gamma if hasattr(blk, "ls2") and hasattr(blk.ls2, "gamma") else None
        if (g1 is None or g2 is None) and one is None:
            one = torch.ones(c, device=x.device)
        if own["qkv"]:
            qkv = ops.tok_linear(x, blk.attn.qkv.weight, blk.attn.qkv.bias, ln=(blk.norm1.weight, blk.norm1.bias, blk.norm1.eps))
        else:
            qkv = blk.attn.qkv(blk.norm1(x))
        if own["attn"] and n <= ops.TOK_ATTENTION_MAX_TOKENS:
            y = ops.tok_attention(qkv, b, n, h)
        else:
            q, k, v = qkv.view(b, n, 3, h, c // h).permute(2, 0, 3, 1, 4).unbind(0)
            y = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(m, c)
        if own["proj"]:
            ops.tok_linear(y, blk.attn.proj.weight, blk.attn.proj.bias, epilogue=ops.TOK_EPI_RESID, residual=x, gamma=g1, out=x)
        else:
            x = torch.addcmul(x, blk.attn.proj(y), g1 if g1 is not None else one)
        if own["fc1"]:
            hid = ops.tok_linear(x, blk.mlp.fc1.weight, blk.mlp.fc1.bias, ln=(blk.norm2.weight, blk.norm2.bias, blk.norm2.eps), epilogue=ops.TOK_EPI_GELU)
        else:
            hid = F.gelu(blk.mlp.fc1(blk.norm2(x)))
        if own["fc2"]:
            ops.tok_linear(hid, blk.mlp.fc2.weight, blk.mlp.fc2.bias, epilogue=ops.TOK_EPI_RESID, residual=x, gamma=g2, out=x)
        else:
            x = torch.addcmul(x, blk.mlp.fc2(hid), g2 if g2 is not None else one)
    return x.view(b, n, c)


def create_backbone(type="dino", backbone: Optional[torch.nn.Module] = None, **kwargs):
    """(model, wh, num_features) as backbone.py:6-22."""
    if type != "dino":
        raise NotImplementedError("only the 'dino' backbone is on the accelerated path (reference default, "
                                  "pretrain_eval_attention.py:54)")
    if backbone is None:
        try:
            backbone = torch.hub.load("facebookresearch/dinov2", "dinov2_vits14")
            proj = getattr(getattr(backbone, "patch_embed", None), "proj", None)
            if isinstance(proj, torch.nn.Conv2d):
                proj.__class__ = _PatchConv          # same parameters, GEMM evaluation
        except Exception as e:  # offline: no network, no cache
            # The reference fails here (backbone.py:15).  A randomly initialised ViT gives meaningless tokens, so it is only
            # substituted on explicit request (tests, bench.py and smoke() set the variable: no weights can be downloaded there).
            if os.environ.get("SIXDGS_RANDOM_BACKBONE", "0") != "1":
                raise RuntimeError(f"6dgs_amd: DINOv2 ViT-S/14 weights unavailable ({e.__class__.__name__}: {e}); pass backbone=<module> "
                                   "or set SIXDGS_RANDOM_BACKBONE=1 to run with a randomly initialised ViT-S/14 (benchmarks/tests only)") from e
            warnings.warn(f"DINOv2 weights unavailable ({e.__class__.__name__}); SIXDGS_RANDOM_BACKBONE=1: randomly initialised ViT-S/14")
            backbone = ViTS14()
    return backbone, (16, 16), 384


def _resize_short_side(x, size, mode):
    h, w = x.shape[-2:]
    if h <= w:
        nh, nw = size, int(size * w / h)
    else:
        nh, nw = int(size * h / w), size
    if (nh, nw) == (h, w):
        return x
    return F.interpolate(x, size=(nh, nw), mode=mode, align_corners=False, antialias=True)


def _center_crop(x, size):
    h, w = x.shape[-2:]
    top, left = int(round((h - size) / 2.0)), int(round((w - size) / 2.0))
    return x[..., top:top + size, left:left + size]


class BatchedTokens:
    """The [n, T, 384 + 14] token block of n images that keep all T tokens, held as its two parts: patch features [n,T,384] and
    the grid position encoding [T,14] shared by all images.  Behaves like the dense tensor where the path reads it (shape, len,
    indexing an image, .dense())."""

    def __init__(self, feats: torch.Tensor, pe: torch.Tensor):
        self.feats, self.pe = feats, pe
        self.shape = torch.Size((feats.shape[0], feats.shape[1], feats.shape[2] + pe.shape[1]))
        self.device = feats.device

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, i):
        if isinstance(i, int):
            return torch.cat([self.feats[i], self.pe], dim=-1)
        return BatchedTokens(self.feats[i], self.pe)

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def dense(self) -> torch.Tensor:
        return torch.cat([self.feats, self.pe[None].expand(self.shape[0], -1, -1)], dim=-1)


class BackboneWrapper(torch.nn.Module):
    def __init__(self, backbone_type: str = "dino", backbone: Optional[torch.nn.Module] = None) -> None:
        super().__init__()
        assert backbone_type in ["dino", "superpoint"]
        self.image_preprocessing_net, backbone_wh, img_num_features = create_backbone(type=backbone_type, backbone=backbone)
        self.norm_mean = torch.nn.Parameter(torch.tensor(IMAGENET_DEFAULT_MEAN, dtype=torch.float32), requires_grad=False)
        self.norm_std = torch.nn.Parameter(torch.tensor(IMAGENET_DEFAULT_STD, dtype=torch.float32), requires_grad=False)
        self.backbone_wh = backbone_wh
        self.img_num_features = img_num_features
        self._pe_cache = {}

    # -- the two transform pipelines of backbone.py:52-77 ----------------------------------------------
    def transformations(self, x):
        x = _center_crop(_resize_short_side(x, 256, "bicubic"), 224)
        key = ("norm", x.dtype, str(x.device))          # device constants are built once (no H2D copy per call: the path
        if key not in self._pe_cache:                   # stays capturable in a hipGraph)
            self._pe_cache[key] = (torch.tensor(IMAGENET_DEFAULT_MEAN, dtype=x.dtype, device=x.device).view(1, 3, 1, 1),
                                   torch.tensor(IMAGENET_DEFAULT_STD, dtype=x.dtype, device=x.device).view(1, 3, 1, 1))
        mean, std = self._pe_cache[key]
        return (x - mean) / std

    def mask_transformations(self, m):
        m = _center_crop(_resize_short_side(m, 256, "bilinear"), 224)
        return F.interpolate(m, size=(self.backbone_wh[0], self.backbone_wh[0]), mode="bilinear", align_corners=False,
                             antialias=True)

    def preprocess(self, img, mask):
        """img [H,W,3] fp32, mask [H,W] bool (or None = all pixels valid) -> (norm_img [1,3,224,224], token mask [16,16]
        bool or None).  An all-ones mask stays all-ones through the bilinear resizes (> 0.1 everywhere, backbone.py:88),
        so the three resize kernels are skipped for RGB images."""
        norm_img = self.transformations(img[None].permute(0, 3, 1, 2))
        mask_img = None if mask is None else self.mask_transformations(mask[None, None] * 1.0)[0, 0] > 0.1
        return norm_img, mask_img

    def preprocess_batch(self, imgs):
        """[n,H,W,3] fp32 (same size, no alpha) -> [n,3,224,224]"""
        return self.transformations(imgs.permute(0, 3, 1, 2))

    def preprocess_batch_u8(self, imgs_u8, table256):
        """[n,H,W,3] uint8 on the GPU (same size, no alpha) -> [n,3,224,224] through ONE kernel (ops.image_prep: table lookup, antialiased bicubic resize, crop,
        normalisation -- the arithmetic of `transformations` on `table256[value]`), or None where it does not apply (the caller then converts and calls
        preprocess_batch)."""
        from . import ops
        if not (ops.image_prep_enabled() and imgs_u8.is_cuda and imgs_u8.dtype == torch.uint8):
            return None
        return ops.image_prep(imgs_u8, table256, IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD, 256, 224)

    def position_encoding(self, dtype, device):
        key = (dtype, str(device))
        if key not in self._pe_cache:
            self._pe_cache[key] = self.get_img_position_encoding(self.backbone_wh, 3, dtype=dtype, device=device)
        return self._pe_cache[key]

    def features_from_norm(self, norm_imgs):
        """[B,3,224,224] -> patch tokens [B,16,16,384]"""
        tok = self.image_preprocessing_net.forward_features(norm_imgs)["x_norm_patchtokens"]
        return tok.reshape(tok.shape[0], self.backbone_wh[0], self.backbone_wh[1], self.img_num_features)

    def assemble(self, feat_hw, mask_img):
        """one image: feat_hw [16,16,384], mask [16,16] (None = all) -> (tokens+pe [T,398], tokens [T,384], fmap [384,16,16])"""
        pe = self.position_encoding(feat_hw.dtype, feat_hw.device)
        with_pe = torch.cat([feat_hw, pe], dim=-1)
        if mask_img is None:
            return with_pe.reshape(-1, with_pe.shape[-1]), feat_hw.reshape(-1, feat_hw.shape[-1]), feat_hw.permute(2, 0, 1)
        return with_pe[mask_img].view(-1, with_pe.shape[-1]), feat_hw[mask_img].view(-1, feat_hw.shape[-1]), feat_hw.permute(2, 0, 1)

    def assemble_batch(self, feats):
        """[n,16,16,384] (all tokens valid) -> (tokens+pe as BatchedTokens [n,256,398], fmaps [n,384,16,16]).  The 14 position
        channels are the same for every image, so the concatenation (backbone.py:110-114) is not materialised: the scorer's
        q_proj takes the patch features and adds the position part of the projection as a per-token bias."""
        pe = self.position_encoding(feats.dtype, feats.device)
        n = feats.shape[0]
        return BatchedTokens(feats.reshape(n, -1, feats.shape[-1]), pe.reshape(-1, pe.shape[-1])), feats.permute(0, 3, 1, 2)

    def forward(self, img, mask=None):
        norm_img, mask_img = self.preprocess(img, mask)
        feat = self.features_from_norm(norm_img)[0]
        return self.assemble(feat, mask_img)

    @staticmethod
    def get_img_position_encoding(img_features_shape, freqs, dtype=torch.float32, device="cpu"):
        """The 2 + 4*freqs grid channels appended to every patch token (semantics of backbone.py:116-139): per cell of the
        h x w token grid its (row, col) coordinate u in [-1,1]^2, then sin(u_a * 2^f) for axis-major / octave-minor (a, f),
        then the cosines in the same order.  Built as one outer product per axis and broadcast over the grid -- the table is
        tiny (16 x 16 x 14) and cached per (dtype, device) by position_encoding()."""
        h, w = img_features_shape
        octave = torch.exp2(torch.arange(freqs, dtype=torch.float32, device=device))
        axis_u = (torch.linspace(-1.0, 1.0, steps=h, dtype=dtype, device=device), torch.linspace(-1.0, 1.0, steps=w, dtype=dtype, device=device))
        out = torch.empty(h, w, 2 + 4 * freqs, dtype=dtype, device=device)
        for a, u in enumerate(axis_u):
            view = (h, 1) if a == 0 else (1, w)
            out[..., a] = u.view(*view)
            phase = (u[:, None] * octave).view(*view, freqs)                    # [h,1,F] or [1,w,F]
            out[..., 2 + a * freqs: 2 + (a + 1) * freqs] = torch.sin(phase)
            out[..., 2 + (2 + a) * freqs: 2 + (3 + a) * freqs] = torch.cos(phase)
        return out
