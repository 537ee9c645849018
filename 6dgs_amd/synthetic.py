"""Deterministic synthetic inputs for the 6DGS pose path (numpy only, no torch RNG).

Everything the parity fixtures, the tests and ``bench.py`` feed to the path comes from
here so that the golden generator (which runs the reference in the build container)
and the GPU box see bit-identical inputs.  Distributions follow SURVEY.md §8(d):

* Gaussians: ``xyz ~ N(0, I3)``, ``scale = 0.005 + 0.05 U(0,1)`` per axis (stored as
  log, the 3DGS convention, reference ``scene/gaussian_model.py:125-127``),
  ``rot ~ N(0, I4)`` (normalised on use, ``gaussian_model.py:129-131``),
  ``f_dc ~ 0.3 N``, ``f_rest ~ 0.05 N`` (16 SH coefficients x 3 channels).
* Scorer weights: the shapes of ``id_module.th["model_state_dict"]`` with
  PyTorch-default-like initialisation (U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for Linear
  / Conv, Xavier-uniform + zero bias for q_proj/k_proj as
  ``our_multihead_attention.py:62-68`` does).
* Image tokens: ``N(0,1)`` [T, 398] when the backbone is excluded.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np

F32 = np.float32


def make_scene(n: int, seed: int = 0, sh_degree: int = 3, scale_lo: float = 0.005,
               scale_span: float = 0.05) -> Dict[str, np.ndarray]:
    """Synthetic 3DGS scene as the raw (pre-activation) parameter arrays."""
    rng = np.random.default_rng(seed)
    xyz = rng.standard_normal((n, 3)).astype(F32)
    scale = (scale_lo + scale_span * rng.random((n, 3))).astype(F32)
    log_scale = np.log(scale).astype(F32)
    rot = rng.standard_normal((n, 4)).astype(F32)
    n_rest = (sh_degree + 1) ** 2 - 1
    f_dc = (0.3 * rng.standard_normal((n, 1, 3))).astype(F32)
    f_rest = (0.05 * rng.standard_normal((n, n_rest, 3))).astype(F32)
    opacity = rng.standard_normal((n, 1)).astype(F32)
    return {
        "xyz": xyz,
        "log_scale": log_scale,
        "rot": rot,
        "f_dc": f_dc,
        "f_rest": f_rest,
        "opacity": opacity,
        "sh_degree": np.int64(sh_degree),
    }


def _uniform(rng, shape, bound):
    return ((rng.random(shape) * 2.0 - 1.0) * bound).astype(F32)


def _linear(rng, out_f, in_f):
    b = 1.0 / math.sqrt(in_f)
    return _uniform(rng, (out_f, in_f), b), _uniform(rng, (out_f,), b)


def _xavier(rng, out_f, in_f):
    b = math.sqrt(6.0 / (in_f + out_f))
    return _uniform(rng, (out_f, in_f), b), np.zeros((out_f,), F32)


def _conv(rng, ch, k):
    b = 1.0 / math.sqrt(ch * k * k)
    return _uniform(rng, (ch, ch, k, k), b), _uniform(rng, (ch,), b)


def make_scorer_state_dict(seed: int = 0, with_cnn: bool = False) -> Dict[str, np.ndarray]:
    """State dict with the key names/shapes of the reference ``IdentificationModule``
    (``identification_module.py:13-46``; keys listed in SURVEY.md §8(b))."""
    rng = np.random.default_rng(1000 + seed)
    sd: Dict[str, np.ndarray] = {}
    for name, (o, i) in (
        ("ray_preprocessor.mlp.0", (512, 141)),
        ("ray_preprocessor.mlp.2", (512, 512)),
        ("ray_preprocessor.mlp2.0", (512, 653)),
        ("ray_preprocessor.mlp2.2", (384, 512)),
    ):
        w, b = _linear(rng, o, i)
        sd[name + ".weight"], sd[name + ".bias"] = w, b
    for name, (o, i) in (("attention.q_proj", (384, 398)), ("attention.k_proj", (384, 384))):
        w, b = _xavier(rng, o, i)
        sd[name + ".weight"], sd[name + ".bias"] = w, b
    if with_cnn:
        rng2 = np.random.default_rng(2000 + seed)
        for name, k in (
            ("camera_direction_prediction_network.dim_reducer1.0", 5),
            ("camera_direction_prediction_network.dim_reducer1.2", 5),
            ("camera_direction_prediction_network.dim_reducer1.4", 5),
            ("camera_direction_prediction_network.dim_reducer2.0", 4),
        ):
            w, b = _conv(rng2, 384, k)
            sd[name + ".weight"], sd[name + ".bias"] = w, b
        for name, (o, i) in (
            ("camera_direction_prediction_network.mlp.0", (256, 384)),
            ("camera_direction_prediction_network.mlp.2", (3, 256)),
        ):
            w, b = _linear(rng2, o, i)
            sd[name + ".weight"], sd[name + ".bias"] = w, b
        sd["backbone_wrapper.norm_mean"] = np.array([0.485, 0.456, 0.406], F32)
        sd["backbone_wrapper.norm_std"] = np.array([0.229, 0.224, 0.225], F32)
    return sd


def make_tokens(t: int, seed: int = 0, scale: float = 1.0, dim: int = 398) -> np.ndarray:
    """Stand-in for ``features_img_w_pe_flat`` [T, 398] (``backbone.py:110-114``)."""
    rng = np.random.default_rng(3000 + seed)
    return (scale * rng.standard_normal((t, dim))).astype(F32)


def make_rays(r: int, seed: int = 0) -> Dict[str, np.ndarray]:
    """Free-standing ray set (origins ~ N(0,1), unit directions, rgb in [0,1.2))."""
    rng = np.random.default_rng(4000 + seed)
    ori = rng.standard_normal((r, 3)).astype(F32)
    d = rng.standard_normal((r, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    rgb = (1.2 * rng.random((r, 3))).astype(F32)
    return {"ori": ori, "dir": d.astype(F32), "rgb": rgb}


def random_rotation(rng) -> np.ndarray:
    q, r = np.linalg.qr(rng.standard_normal((3, 3)))
    q = q * np.sign(np.diag(r))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def make_cameras(b: int, seed: int = 0, width: int = 800, height: int = 800, fov: float = 0.8,
                 rgba: bool = False):
    """``b`` synthetic query views: random rotation, T=(0,0,4), uint8 U{0..255} image.
    Returned as plain dicts with the ``CameraInfo`` field names (``scene_structure.py:7-17``)."""
    rng = np.random.default_rng(5000 + seed)
    cams = []
    for i in range(b):
        rot = random_rotation(rng)
        ch = 4 if rgba else 3
        img = rng.integers(0, 256, size=(height, width, ch), dtype=np.uint8)
        cams.append({
            "uid": i, "R": rot.astype(np.float64), "T": np.array([0.0, 0.0, 4.0]),
            "FovY": fov, "FovX": fov, "image": img, "image_path": "", "image_name": f"syn_{i}",
            "width": width, "height": height,
        })
    return cams


def make_masked_cameras(seed: int = 9, size: int = 120):
    """RGBA query views whose alpha channel has STRUCTURE (a disc, a half plane with a soft edge, a small off-centre box), so
    that the mask -> token selection of the backbone wrapper (backbone.py:86-114) keeps a proper subset of the 256 tokens."""
    cams = make_cameras(3, seed, width=size, height=size, rgba=True)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float64)
    c = (size - 1) / 2.0
    disc = ((yy - c) ** 2 + (xx - c) ** 2) <= (0.33 * size) ** 2
    ramp = np.clip((xx - 0.45 * size) / (0.2 * size), 0.0, 1.0)
    box = (yy > 0.1 * size) & (yy < 0.45 * size) & (xx > 0.55 * size) & (xx < 0.95 * size)
    for cam, a in zip(cams, (disc * 255.0, ramp * 255.0, box * 200.0)):
        cam["image"] = cam["image"].copy()
        cam["image"][..., 3] = a.astype(np.uint8)
    return cams


def write_dataset_fixtures(root: str, seed: int = 0, n_views: int = 10, width: int = 20, height: int = 14):
    """Tiny on-disk scenes in the four layouts the reference's loaders read (scene/dataset_loader.py): COLMAP binary and text models
    (Mip-NeRF360 layout), NSVF / Tanks&Temples, Blender -- plus an experiment directory (`<prefix>_<id>/point_cloud/iteration_N/
    point_cloud.ply` + `cfg_args`) for each.  Deterministic (numpy RNG + PIL PNG encoder).  Returns {name: source_path}."""
    import json
    import os
    import struct
    from PIL import Image

    rng = np.random.default_rng(7000 + seed)
    out = {}

    def poses(n):
        ps = []
        for _ in range(n):
            c2w = np.eye(4)
            c2w[:3, :3] = random_rotation(rng)
            c2w[:3, 3] = rng.standard_normal(3) * 2.0
            ps.append(c2w)
        return ps

    def quat(rot):       # rotation matrix -> COLMAP (w, x, y, z), w >= 0
        w = math.sqrt(max(0.0, 1.0 + rot[0, 0] + rot[1, 1] + rot[2, 2])) / 2.0
        x = math.copysign(math.sqrt(max(0.0, 1.0 + rot[0, 0] - rot[1, 1] - rot[2, 2])) / 2.0, rot[2, 1] - rot[1, 2])
        y = math.copysign(math.sqrt(max(0.0, 1.0 - rot[0, 0] + rot[1, 1] - rot[2, 2])) / 2.0, rot[0, 2] - rot[2, 0])
        z = math.copysign(math.sqrt(max(0.0, 1.0 - rot[0, 0] - rot[1, 1] + rot[2, 2])) / 2.0, rot[1, 0] - rot[0, 1])
        return np.array([w, x, y, z])

    # ---- COLMAP, binary + text (two cameras: PINHOLE and SIMPLE_PINHOLE in the binary model; PINHOLE only in the text model)
    for kind in ("colmap_bin", "colmap_txt"):
        src = os.path.join(root, kind)
        os.makedirs(os.path.join(src, "sparse", "0"), exist_ok=True)
        os.makedirs(os.path.join(src, "images"), exist_ok=True)
        open(os.path.join(src, "sparse", "0", "points3D.ply"), "wb").close()       # present (unreadable): the loaders do not regenerate it
        cams = [(1, 1, width, height, [17.5, 16.25, width / 2.0, height / 2.0])]
        if kind == "colmap_bin":
            cams.append((2, 0, width, height, [19.0, width / 2.0, height / 2.0]))
        names = [f"img_{(i * 7) % n_views:03d}.png" for i in range(n_views)]           # file order != name order
        recs = []
        for i, (nm, c2w) in enumerate(zip(names, poses(n_views))):
            w2c = np.linalg.inv(c2w)
            recs.append((i + 1, quat(w2c[:3, :3]), w2c[:3, 3], cams[i % len(cams)][0], nm))
            Image.fromarray(rng.integers(0, 256, size=(height, width, 3), dtype=np.uint8), "RGB").save(os.path.join(src, "images", nm))
        if kind == "colmap_bin":
            with open(os.path.join(src, "sparse", "0", "cameras.bin"), "wb") as f:
                f.write(struct.pack("<Q", len(cams)))
                for cid, model, w, h, par in cams:
                    f.write(struct.pack("<iiQQ", cid, model, w, h) + struct.pack("<" + "d" * len(par), *par))
            with open(os.path.join(src, "sparse", "0", "images.bin"), "wb") as f:
                f.write(struct.pack("<Q", len(recs)))
                for iid, q, t, cid, nm in recs:
                    f.write(struct.pack("<idddddddi", iid, *q, *t, cid) + nm.encode() + b"\x00")
                    npts = int(rng.integers(0, 4))
                    f.write(struct.pack("<Q", npts))
                    for _ in range(npts):
                        f.write(struct.pack("<ddq", float(rng.random()), float(rng.random()), -1))
        else:
            with open(os.path.join(src, "sparse", "0", "cameras.txt"), "w") as f:
                f.write("# Camera list with one line of data per camera:\n")
                for cid, model, w, h, par in cams:
                    f.write(f"{cid} PINHOLE {w} {h} " + " ".join(repr(float(v)) for v in par) + "\n")
            with open(os.path.join(src, "sparse", "0", "images.txt"), "w") as f:
                f.write("# Image list with two lines of data per image:\n")
                for iid, q, t, cid, nm in recs:
                    f.write(f"{iid} " + " ".join(repr(float(v)) for v in list(q) + list(t)) + f" {cid} {nm}\n")
                    f.write("1.0 2.0 -1\n")
        out[kind] = src

    # ---- NSVF / Tanks&Temples: split 0 = train, 1 = test here (no split 2: the loader falls back to 1), RGBA images
    src = os.path.join(root, "tt")
    os.makedirs(os.path.join(src, "pose"), exist_ok=True)
    os.makedirs(os.path.join(src, "rgb"), exist_ok=True)
    open(os.path.join(src, "points3d.ply"), "wb").close()
    np.savetxt(os.path.join(src, "intrinsics.txt"), np.array([[21.0, 0, width / 2.0, 0], [0, 20.5, height / 2.0, 0], [0, 0, 1, 0], [0, 0, 0, 1]]))
    for i, c2w in enumerate(poses(n_views)):
        tag = f"{0 if i % 3 else 1}_{i:04d}"
        np.savetxt(os.path.join(src, "pose", tag + ".txt"), c2w)
        Image.fromarray(rng.integers(0, 256, size=(height, width, 4), dtype=np.uint8), "RGBA").save(os.path.join(src, "rgb", tag + ".png"))
    out["tt"] = src

    # ---- Blender
    src = os.path.join(root, "blender")
    for split, n in (("train", n_views - 3), ("test", 3)):
        os.makedirs(os.path.join(src, split), exist_ok=True)
        frames = []
        for i, c2w in enumerate(poses(n)):
            Image.fromarray(rng.integers(0, 256, size=(height, width, 4), dtype=np.uint8), "RGBA").save(os.path.join(src, split, f"r_{i}.png"))
            frames.append({"file_path": f"./{split}/r_{i}", "transform_matrix": c2w.tolist()})
        with open(os.path.join(src, f"transforms_{split}.json"), "w") as f:
            json.dump({"camera_angle_x": 0.69, "frames": frames}, f)
    open(os.path.join(src, "points3d.ply"), "wb").close()
    out["blender"] = src
    return out


def checksum(sd: Dict[str, np.ndarray]) -> int:
    """Exact (integer, order-independent) checksum of a dict of fp32 arrays -- guards the fixtures against a
    drift of numpy's RNG streams.  Pure integer arithmetic, so it is identical on every host."""
    tot = 0
    for k in sorted(sd):
        bits = np.frombuffer(np.ascontiguousarray(sd[k], dtype=np.float32).tobytes(), dtype=np.uint32).astype(np.uint64)
        w = (np.arange(bits.size, dtype=np.uint64) % np.uint64(65521)) + np.uint64(1)
        tot = (tot + int((bits * w).sum(dtype=np.uint64))) % (1 << 61)
    return tot
