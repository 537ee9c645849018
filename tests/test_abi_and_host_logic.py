"""CPU tests: the C-ABI library loads and exports every symbol include/sixdgs.h declares (no compute
calls without a GPU), the product path refuses to run without the GPU, and the host-side logic
(token padding, image prep, sharding, results schema helpers) behaves."""
import ctypes as C
import importlib
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "sixdgs.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sixdgs_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    b = importlib.import_module("6dgs_amd.build")
    lib = C.CDLL(b.build())
    syms = header_symbols()
    assert len(syms) >= 28
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/sixdgs.h but not exported"
    L = importlib.import_module("6dgs_amd._lib")
    assert set(L.SIGNATURES) == set(syms), set(L.SIGNATURES) ^ set(syms)
    ge = importlib.import_module("__graft_entry__")
    assert L.load().sixdgs_abi_version() == L.ABI_VERSION == ge.header_abi_version()
    ge.post_build_checks()       # the checks build() ends with: they must pass on the built tree (round-1 bug: a stale literal)
    # argument errors are reported without touching the GPU
    assert L.load().sixdgs_mask_degraded(None, -1, 50, None, None) == -1
    assert b"bad argument" in L.load().sixdgs_error_string(-1)
    assert L.load().sixdgs_packed_weights_floats() > 1_000_000
    assert L.load().sixdgs_score_topk_workspace_bytes(32_000_000, 1, 100) > 32_000_000 * 1024


def test_product_path_has_no_cpu_fallback(syn):
    pkg = importlib.import_module("6dgs_amd")
    ops = importlib.import_module("6dgs_amd.ops")
    scene = pkg.GaussianScene.from_dict(syn.make_scene(10, 0), device="cpu")
    with pytest.raises(RuntimeError):
        pkg.generate_all_possible_rays(scene)
    with pytest.raises(RuntimeError):
        ops.mask_degraded(torch.zeros(4, 3))
    with pytest.raises(RuntimeError):
        pkg.test_pose_estimation([], None.__class__ and pkg.IdentificationModule("dino"), torch.zeros(4, 3), torch.zeros(4, 3),
                                 torch.zeros(4, 3), torch.tensor([0.0, 1.0, 0.0]), verbose=False)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "6dgs_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("the oracle", "").replace("CPU oracle", "").replace("like the oracle", "") \
                    or f in ("hostcheck.cpp", "device_math.h", "pose.hip", "__init__.py"), f
                assert "import oracle" not in txt and "from oracle" not in txt and "sixdgs_oracle" not in txt, f


def test_state_dict_keys_match_reference(syn):
    pkg = importlib.import_module("6dgs_amd")
    idm = pkg.IdentificationModule("dino")
    sd = syn.make_scorer_state_dict(0, with_cnn=True)
    res = idm.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not res.unexpected_keys
    assert all(k.startswith("backbone_wrapper.image_preprocessing_net") for k in res.missing_keys)
    assert idm.state_dict()["ray_preprocessor.mlp2.0.weight"].shape == (512, 653)
    assert idm.state_dict()["attention.q_proj.weight"].shape == (384, 398)
    # the training forward is plain PyTorch (autograd): it also runs on CPU tensors
    rays = syn.make_rays(300, 1)
    o, d, c = (torch.from_numpy(rays[k]) for k in ("ori", "dir", "rgb"))
    torch.manual_seed(0)
    scores, att, feat, up, used = idm(torch.rand(40, 40, 3), torch.ones(40, 40, dtype=torch.bool), o, d, c, rays_to_test=200)
    assert scores.shape == (200,) and att.shape == (256, 200) and feat.shape == (256, 384) and up.shape == (3,) and used.shape == (200,)
    assert abs(float(scores.sum()) - 256.0) < 1e-2 and abs(float(up.norm()) - 1.0) < 1e-5
    (scores.square().sum() + up[0]).backward()
    assert idm.attention.q_proj.weight.grad is not None and idm.ray_preprocessor.mlp[0].weight.grad.abs().sum() > 0
    assert idm.camera_direction_prediction_network.mlp[2].weight.grad.abs().sum() > 0


def test_image_prep_and_gt_pose_on_cpu(syn):
    tp = importlib.import_module("6dgs_amd.test")
    pkg = importlib.import_module("6dgs_amd")
    cam = pkg.CameraInfo(**syn.make_cameras(1, 1, 32, 24, rgba=True)[0])
    img, mask = tp.prepare_image(cam.image, "cpu")
    a = np.asarray(cam.image).astype(np.float32) / np.float32(255)
    assert np.abs(img.numpy() - (a[..., :3] * a[..., 3:] + (1 - a[..., 3:]))).max() == 0
    assert (mask.numpy() == (a[..., 3] > 0.3)).all()
    c2w, K = tp.gt_pose_and_intrinsics(cam, "cpu")
    w2c = np.eye(4)
    w2c[:3, :3] = cam.R.T
    w2c[:3, 3] = cam.T
    assert np.abs(c2w.numpy() - np.linalg.inv(w2c)).max() < 1e-5
    assert abs(K[0, 0].item() - 32 / (2 * np.tan(0.4))) < 1e-3 and K[0, 2].item() == 16


def test_backbone_wrapper_shapes_cpu():
    bb = importlib.import_module("6dgs_amd.backbone")
    w = bb.BackboneWrapper("dino", backbone=bb.ViTS14(depth=1))
    img = torch.rand(60, 80, 3)
    mask = torch.zeros(60, 80, dtype=torch.bool)
    mask[:, 20:60] = True
    with torch.no_grad():
        t_pe, t, fmap = w(img, mask)
    assert t_pe.shape[1] == 398 and t.shape[1] == 384 and fmap.shape == (384, 16, 16)
    assert 0 < t_pe.shape[0] < 256 and t_pe.shape[0] == t.shape[0]
    pe = w.get_img_position_encoding((16, 16), 3)
    assert pe.shape == (16, 16, 14) and pe[0, 0, 0] == -1 and pe[15, 15, 1] == 1


def test_pad_tokens_cpu():
    ops = importlib.import_module("6dgs_amd.ops")
    tok, n = ops.pad_tokens([torch.ones(3, 398), torch.ones(0, 398), torch.ones(256, 398)], "cpu")
    assert tok.shape == (3, 256, 398) and n.tolist() == [3, 0, 256] and tok[0, 3:].abs().sum() == 0
    with pytest.raises(RuntimeError):
        ops.pad_tokens([torch.ones(257, 398)], "cpu")


def test_shard_range_covers_everything():
    dd = importlib.import_module("6dgs_amd.distributed")
    for n in (0, 1, 7, 64, 128, 1001):
        for world in (1, 2, 3, 8):
            spans = [dd.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_ply_ingestion_follows_the_reference_layout(tmp_path):
    """A file written the way gaussian_model.py:298-334 writes it (independent numpy writer: structured dtype in the order of
    construct_list_of_attributes, f_dc / f_rest channel-major) loads into the arrays gaussian_model.py:342-420 builds;
    save_ply writes the same bytes back; double-precision and shuffled property orders load too."""
    import importlib
    pkg = importlib.import_module("6dgs_amd")
    rng = np.random.default_rng(0)
    n, deg = 257, 3
    xyz, nrm = rng.standard_normal((n, 3)).astype("f4"), np.zeros((n, 3), "f4")
    f_dc_file = rng.standard_normal((n, 3)).astype("f4")               # f_dc_c   = channel c
    f_rest_file = rng.standard_normal((n, 45)).astype("f4")             # f_rest_j = channel j // 15, coefficient j % 15
    op, sc, rot = rng.standard_normal((n, 1)).astype("f4"), rng.standard_normal((n, 3)).astype("f4"), rng.standard_normal((n, 4)).astype("f4")
    names = (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(45)] + ["opacity"]
             + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)])
    attrs = np.concatenate([xyz, nrm, f_dc_file, f_rest_file, op, sc, rot], axis=1)

    def write(path, names, attrs, ftype="float", dt="<f4"):
        el = np.empty(n, dtype=[(nm, dt) for nm in names])
        for j, nm in enumerate(names):
            el[nm] = attrs[:, j]
        with open(path, "wb") as f:
            f.write(b"ply\nformat binary_little_endian 1.0\ncomment written by the test\n")
            f.write(f"element vertex {n}\n".encode() + "".join(f"property {ftype} {nm}\n" for nm in names).encode() + b"end_header\n")
            f.write(el.tobytes())

    p1 = str(tmp_path / "point_cloud.ply")
    write(p1, names, attrs)
    s = pkg.GaussianScene.load_ply(p1, deg, device="cpu")
    assert len(s) == n and s.active_sh_degree == 3
    assert np.array_equal(s._xyz.numpy(), xyz) and np.array_equal(s._opacity.numpy(), op)
    assert np.array_equal(s._scaling.numpy(), sc) and np.array_equal(s._rotation.numpy(), rot)
    assert s._features_dc.shape == (n, 1, 3) and np.array_equal(s._features_dc.numpy()[:, 0, :], f_dc_file)
    assert s._features_rest.shape == (n, 15, 3)
    assert np.array_equal(s._features_rest.numpy(), f_rest_file.reshape(n, 3, 15).transpose(0, 2, 1))
    p2 = str(tmp_path / "out" / "point_cloud.ply")
    s.save_ply(p2)
    hdr_end = open(p2, "rb").read().index(b"end_header\n") + 11
    assert open(p2, "rb").read()[hdr_end:] == attrs.astype("<f4").tobytes()
    perm = rng.permutation(len(names))                                   # any property order, doubles
    p3 = str(tmp_path / "shuffled.ply")
    write(p3, [names[i] for i in perm], attrs[:, perm], "double", "<f8")
    s3 = pkg.GaussianScene.load_ply(p3, deg, device="cpu")
    for k in ("_xyz", "_scaling", "_rotation", "_features_dc", "_features_rest", "_opacity"):
        assert torch.equal(getattr(s, k), getattr(s3, k)), k
    with pytest.raises(RuntimeError):
        pkg.GaussianScene.load_ply(p1, 2, device="cpu")                    # wrong SH degree (gaussian_model.py:368)


def test_bench_plain_multi_gpu_start_becomes_the_launcher():
    """`python bench.py --gpus 2` without a rank environment must re-launch itself under torch.distributed.run (a plain
    start used to SystemExit before any rank existed).  No GPU here: the two ranks must get as far as the "needs an MI355X" exit,
    which proves the launcher ran and handed RANK / WORLD_SIZE to bench.py."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SIXDGS_BENCH_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, "-W", "ignore", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--gaussians", "100"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_bench_contract.py")
    assert p.returncode != 0
    assert "needs an MI355X" in p.stderr and "WORLD_SIZE=1" not in p.stderr, p.stderr[-1500:]


def test_select_sample_indices_and_batched_tokens_host_logic():
    """Host-side pieces of the select path and of the fused q_proj that need no GPU: the ray sample (one ray per group of 16 / 32,
    strictly increasing, position inside the group varying from group to group) and the BatchedTokens view."""
    ops = importlib.import_module("6dgs_amd.ops")
    bb = importlib.import_module("6dgs_amd.backbone")
    for r, stride in ((1_000_003, 16), (16_000_000, 32), (40, 16)):
        idx = ops.select_sample_indices(r, "cpu")
        assert idx.shape[0] == r // stride and idx.dtype == torch.int64
        if idx.numel():
            assert bool((idx[1:] > idx[:-1]).all()) and int(idx.max()) < r
            assert torch.equal(idx // stride, torch.arange(idx.shape[0]))                    # exactly one ray of every group
            if idx.numel() > 1000:
                assert len(set((idx % stride).tolist())) == stride                          # every position inside a group is used
                assert bool(((idx % 64)[:4096].reshape(-1, 4).float().std(dim=1) > 0).any())   # not the same iso-cell direction of every ellipsoid
    f, pe = torch.randn(3, 256, 384), torch.randn(256, 14)
    t = bb.BatchedTokens(f, pe)
    assert t.shape == (3, 256, 398) and len(t) == 3 and t.dense().shape == (3, 256, 398)
    assert torch.equal(t[1], torch.cat([f[1], pe], dim=-1)) and torch.equal(t.dense()[2], t[2])
    sub = t[torch.tensor([2, 0])]
    assert isinstance(sub, bb.BatchedTokens) and torch.equal(sub.feats, f[[2, 0]]) and [x.shape for x in t] == [(256, 398)] * 3


def test_dense_kernel_register_budget():
    """k_dense_planes<2,4> sits on the 256-register line: a few more live values and hipcc parks the operand staging registers in scratch, inside
    the slab loop (measured: 2x slower layers).  The resource report of the cross-compile must show every instance (both tile shapes, both
    activation layouts) without scratch, one copy of the slab loop each, and no scratch access between its MFMAs."""
    import subprocess
    import tempfile
    b = importlib.import_module("6dgs_amd.build")
    src = os.path.join(b.CSRC, "dense.hip")
    with tempfile.TemporaryDirectory() as tmp:
        base = [b.HIPCC, *[f for f in b.FLAGS if not f.startswith("-DSDG_")], "--cuda-device-only"]
        rep = subprocess.run(base + ["-c", src, "-o", os.path.join(tmp, "dense.o"), "-Rpass-analysis=kernel-resource-usage"],
                             capture_output=True, text=True, check=True).stderr
        asm = os.path.join(tmp, "dense.s")
        subprocess.run(base + ["-S", src, "-o", asm], capture_output=True, text=True, check=True)
        text = open(asm).read()
    scratch = {}
    for m in re.finditer(r"Function Name: (\S+).*?ScratchSize \[bytes/lane\]: (\d+)", rep, re.S):
        scratch[m.group(1)] = int(m.group(2))
    wide = [v for k, v in scratch.items() if "k_dense_planesILi3ELi2" in k]
    square = [v for k, v in scratch.items() if "k_dense_planesILi2ELi4" in k]
    assert wide == [0, 0, 0], scratch               # <3,2>: ray-major, chunk-major, chunk-major in / key planes out
    assert square == [0, 0], scratch                # <2,4>: ray-major, chunk-major
    names = [k for k in scratch if "k_dense_planes" in k]
    assert len(names) == 5
    for name in names:
        body = text[text.index(name + ":"):]
        body = body[:body.index("s_endpgm")].splitlines()
        mfma = [i for i, l in enumerate(body) if "v_mfma_f32_32x32x16_f16" in l]
        n_per_slab = 48 if "ILi2ELi4" in name else 36
        assert len(mfma) == n_per_slab, (name, len(mfma))                   # one copy of the slab
        inside = [l for l in body[mfma[0]:mfma[-1]] if "scratch_" in l]
        assert not inside, (name, inside[:3])
