"""a23 pinned: BackboneWrapper (resize / crop / normalise, patch tokens, grid position encoding, mask -> token selection)
against the reference's own output, golden g7 (`e2e{i}_tokens`, `e2e{i}_fmap`: captured from the reference's
BackboneWrapper.forward inside test_pose_estimation, oracle/gen_golden.py:g7).  DINOv2 weights do not exist offline, so the
golden run used a fixed patch-embed stand-in (gen_golden.FakeDino: a 14x14 / stride-14 convolution drawn from
np.random.default_rng(77)); the same stand-in is injected here through the `backbone=` argument, which leaves everything
ELSE of backbone.py:82-139 under test.  Plain PyTorch on CPU tensors (the image side stays on PyTorch-ROCm)."""
import importlib

import numpy as np
import pytest
import torch


class PatchEmbedStandIn(torch.nn.Module):
    """Same function as the golden run's stand-in: conv 14x14 stride 14, weights N(0,1)/sqrt(588) from default_rng(77)."""

    def __init__(self):
        super().__init__()
        w = (np.random.default_rng(77).standard_normal((384, 3, 14, 14)) / np.sqrt(3 * 14 * 14)).astype(np.float32)
        self.proj = torch.nn.Conv2d(3, 384, 14, stride=14, bias=False)
        with torch.no_grad():
            self.proj.weight.copy_(torch.from_numpy(w))

    def forward_features(self, x):
        return {"x_norm_patchtokens": self.proj(x).flatten(2).transpose(1, 2)}


@pytest.fixture(scope="module")
def wrapper():
    bb = importlib.import_module("6dgs_amd.backbone")
    return bb.BackboneWrapper("dino", backbone=PatchEmbedStandIn()).eval()


def cameras(syn):
    return syn.make_cameras(3, 7, width=96, height=96, rgba=False) + syn.make_cameras(1, 8, width=80, height=80, rgba=True)


def test_tokens_and_feature_map_match_the_reference(wrapper, syn, golden):
    g = golden("g7_e2e")
    tp = importlib.import_module("6dgs_amd.test")
    assert int(g["e2e_n"]) == 4
    for i, cam in enumerate(cameras(syn)):
        img, mask = tp.prepare_image(cam["image"], "cpu")            # test.py:69-83 (RGBA: composite on white, mask alpha > 0.3)
        with torch.no_grad():
            t_pe, t_flat, fmap = wrapper(img, mask)
        ref_tok, ref_fmap = g[f"e2e{i}_tokens"], g[f"e2e{i}_fmap"]
        assert tuple(t_pe.shape) == ref_tok.shape, (i, t_pe.shape, ref_tok.shape)       # the SAME tokens survive the mask
        assert tuple(fmap.shape) == ref_fmap.shape == (384, 16, 16)
        scale = np.abs(ref_tok[:, :384]).max()
        assert np.abs(t_pe.numpy()[:, :384] - ref_tok[:, :384]).max() / scale < 2e-5      # patch features
        assert np.abs(t_pe.numpy()[:, 384:] - ref_tok[:, 384:]).max() < 1e-6              # 14 position-encoding channels
        assert np.abs(fmap.numpy() - ref_fmap).max() / np.abs(ref_fmap).max() < 2e-5
        assert torch.equal(t_flat, t_pe[:, :384])


def test_mask_to_token_selection_matches_the_reference(wrapper, syn, golden):
    """g7's RGBA camera has noise for alpha (every token survives); golden g11 has structured alpha -- a disc, a soft-edged half
    plane, a small box -- so 140 / 128 / 56 of the 256 tokens survive in the reference (backbone.py:86-114: bilinear mask resizes,
    > 0.1, boolean selection in row-major token order).  Same tokens, same order, same values."""
    g = golden("g11_backbone_masks")
    tp = importlib.import_module("6dgs_amd.test")
    kept = []
    for i, cam in enumerate(syn.make_masked_cameras(9, 120)):
        img, mask = tp.prepare_image(cam["image"], "cpu")
        with torch.no_grad():
            t_pe, t_flat, fmap = wrapper(img, mask)
        ref = g[f"m{i}_tokens"]
        assert tuple(t_pe.shape) == ref.shape, (i, t_pe.shape, ref.shape)
        assert np.abs(t_pe.numpy()[:, 384:] - ref[:, 384:]).max() < 1e-6                     # the grid positions of the survivors
        assert np.abs(t_pe.numpy()[:, :384] - ref[:, :384]).max() / np.abs(ref[:, :384]).max() < 2e-5
        assert np.abs(t_flat.numpy() - g[f"m{i}_flat"]).max() / np.abs(g[f"m{i}_flat"]).max() < 2e-5
        assert np.abs(fmap.numpy() - g[f"m{i}_fmap"]).max() / np.abs(g[f"m{i}_fmap"]).max() < 2e-5
        kept.append(t_pe.shape[0])
    assert kept == [140, 128, 56]


def test_batched_image_side_equals_the_per_image_path(wrapper, syn):
    """IdentificationModule.image_tokens batches same-size RGB images (one launch sequence): identical tokens."""
    tp = importlib.import_module("6dgs_amd.test")
    cams = cameras(syn)[:3]
    imgs = [tp.prepare_image(c["image"], "cpu")[0] for c in cams]
    with torch.no_grad():
        feats = wrapper.features_from_norm(wrapper.preprocess_batch(torch.stack(imgs)))
        toks, fmaps = wrapper.assemble_batch(feats)
        for i, im in enumerate(imgs):
            t_pe, _, fmap = wrapper(im, None)
            assert torch.allclose(toks[i], t_pe, atol=1e-6) and torch.allclose(fmaps[i], fmap, atol=1e-6)
