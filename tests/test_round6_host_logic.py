"""Host-side logic added in round 6 (CPU): arena ownership."""
import importlib

import pytest

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ops():
    return importlib.import_module("6dgs_amd.ops")


def test_arena_serves_one_module_at_a_time(ops):
    """ops.Arena is a bump allocator that an IdentificationModule resets per scene: a reset by a SECOND live module would hand the first one's key planes out
    again under its feet.  reset(owner) claims the arena; another live owner is refused until release_owner() (or the first owner is gone)."""
    class Mod:            # stands in for an IdentificationModule (only identity and lifetime matter)
        pass
    a = ops.Arena(8 << 20, "cpu")
    m1, m2 = Mod(), Mod()
    a.take(1 << 20)
    a.reset(m1)
    assert a.mark() == 0
    a.take(1 << 20)
    a.reset(m1)                                  # the owner may reset as often as it likes
    with pytest.raises(RuntimeError, match="already serves another IdentificationModule"):
        a.reset(m2)
    a.reset()                                    # an anonymous reset (tools, tests) is the caller's responsibility, as before
    a.release_owner()
    a.reset(m2)
    del m2                                       # a dead owner no longer holds the arena
    import gc
    gc.collect()
    a.reset(m1)


def test_taps_major_weight_matches_the_patch_order_of_im2col():
    """camera_direction_network._taps_major_weight: conv.weight with its columns in (kh, kw, c) order, the order ops.im2col(taps_major=True) lays the
    patches out in -- checked on the CPU against F.conv2d with the patch matrix rebuilt from F.unfold (the GPU test checks the kernel against the same
    reordering of unfold)."""
    import importlib
    import torch
    import torch.nn.functional as F
    cdn = importlib.import_module("6dgs_amd.camera_direction_network")
    torch.manual_seed(0)
    net = cdn.CameraDirectionPredictor(image_feature_channel=6, image_size=(16, 16))
    conv = net.dim_reducer1[0]
    x = torch.randn(2, 6, 9, 8)
    k = conv.kernel_size[0]
    ho, wo = x.shape[2] - k + 1, x.shape[3] - k + 1
    cols = F.unfold(x, k)                                                                 # [B, C*k*k, L], rows (c, kh, kw)
    tm = cols.view(2, 6, k * k, ho * wo).permute(0, 3, 2, 1).reshape(2 * ho * wo, k * k * 6)      # rows (b, oy, ox), columns (kh, kw, c)
    wt = net._taps_major_weight(conv)
    assert wt.shape == (conv.out_channels, k * k * 6) and net._taps_major_weight(conv) is wt      # cached per parameter version
    y = (tm @ wt.t() + conv.bias).view(2, ho, wo, -1).permute(0, 3, 1, 2)
    assert torch.allclose(y, conv(x), atol=1e-5)
    with torch.no_grad():
        conv.weight.mul_(2.0)
    assert net._taps_major_weight(conv) is not wt                                         # a changed parameter is permuted again


def test_fused_vit_blocks_are_a_gpu_inference_path_only():
    """backbone.fused_blocks_usable: CPU tensors, gradient mode and SIXDGS_VIT_FUSED=0 keep PyTorch's blocks; every stage has a row-count threshold."""
    import importlib
    import torch
    bb = importlib.import_module("6dgs_amd.backbone")
    vit = bb.ViTS14(depth=1).eval()
    t = torch.zeros(1, 257, 384)
    with torch.no_grad():
        assert not bb.fused_blocks_usable(vit, t)
    assert set(bb.FUSED_MIN_ROWS) == {"qkv", "attn", "proj", "fc1", "fc2"} and all(v >= 0 for v in bb.FUSED_MIN_ROWS.values())
    with torch.no_grad():
        out = vit.forward_features(torch.zeros(1, 3, 224, 224))
    assert out["x_norm_patchtokens"].shape == (1, 256, 384)


def test_image_prep_geometry_is_the_wrappers_resize_and_crop(ops):
    """ops.image_prep_geometry states for sixdgs_image_prep what BackboneWrapper's Resize(256) + CenterCrop(224) do to an h x w image (backbone.py:52-77 as mirrored in
    6dgs_amd/backbone.py): the resized grid and the crop's corner -- checked against the shapes and the content position those two functions produce on the CPU."""
    bb = importlib.import_module("6dgs_amd.backbone")
    for h, w in ((800, 800), (400, 400), (480, 640), (750, 500), (260, 300), (200, 180), (256, 300), (1080, 1920)):
        nh, nw, top, left = ops.image_prep_geometry(h, w)
        x = torch.zeros(1, 1, h, w)
        r = bb._resize_short_side(x, 256, "bicubic")
        assert tuple(r.shape[-2:]) == (nh, nw), (h, w)
        ramp = torch.arange(nh * nw, dtype=torch.float32).view(1, 1, nh, nw)          # which resized-grid pixel lands at the crop's corner
        c = bb._center_crop(ramp, 224)
        assert tuple(c.shape[-2:]) == (224, 224) and int(c[0, 0, 0, 0]) == top * nw + left, (h, w)
    assert ops.image_prep_enabled() in (True, False)
