"""Scene ingestion either side of the pose path: what the reference driver reads before it can call
generate_all_possible_rays / test_pose_estimation on a real scene.

  experiment discovery   parse_exp_dir, get_highest_valid_checkpoint          pose_estimation/file_utils.py:19-72
  training arguments     parse_cfg_args, get_checkpoint_arguments (`cfg_args`) pose_estimation/file_utils.py:13-16, cfg_grammar/
  cameras                COLMAP (Mip-NeRF360), NSVF / Tanks&Temples, Blender   scene/colmap.py:73-131, scene/colmap_utils.py,
                                                                               scene/tanksandtemples.py:110-167, scene/synthetic.py
  dispatch               load_data, get_dataset_prefix                         scene/dataset_loader.py:5-32

Host-side I/O only (numpy + PIL): every loader returns the reference's `SceneInfo(train_cameras, test_cameras, ...)` of
`CameraInfo(uid, R, T, FovY, FovX, image, image_path, image_name, width, height)` tuples, R stored transposed (world-to-camera
rotation transposed, the 3DGS convention), which is exactly what test_pose_estimation consumes.  The sparse point cloud
(`points3D.*`) is only the 3DGS trainer's initialisation and is not read here; `ply_path` is still reported.

The `cfg_args` file is the text of an argparse `Namespace(k=v, ...)`.  The reference parses it with an ANTLR grammar
(cfg_grammar/Namespace.g4: values are INT | FLOAT | BOOL | STRING) and converts booleans with `bool(text)`
(cfg_grammar/parse_config.py:36), so that `eval=False` reads as True -- every scene the reference evaluates is therefore
split train/test.  `parse_cfg_args(..., reference_bools=True)` (the default) reproduces that; pass False for literal values.
"""
from __future__ import annotations

import json
import math
import os
import re
import struct
from typing import Dict, List, NamedTuple, Optional

import numpy as np

from .scene import CameraInfo


class SceneInfo(NamedTuple):          # scene/scene_structure.py:20-25
    point_cloud: object
    train_cameras: List[CameraInfo]
    test_cameras: List[CameraInfo]
    nerf_normalization: dict
    ply_path: str


class dotdict(dict):
    """dict with attribute access; a missing key reads as None (file_utils.py:5-10)."""
    __getattr__ = dict.get
    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__


# ---------------------------------------------------------------------------------------------------------------------------
# cfg_args
# ---------------------------------------------------------------------------------------------------------------------------
_TOKEN = re.compile(r"""\s*(?:
      (?P<bool>True|False|true|false)(?![A-Za-z0-9_])
    | (?P<float>[+-]?[0-9]+\.[0-9]*)
    | (?P<int>[+-]?[0-9]+)
    | (?P<str>'[^'\\\n\r]*'|"[^'\\\n\r]*")
    | (?P<id>[A-Za-z_][A-Za-z0-9_]*)
    | (?P<punct>[=,()])
    )""", re.X)


def parse_cfg_args(text: str, reference_bools: bool = True) -> Dict[str, object]:
    """`Namespace(key=value, ...)` -> dict.  Value kinds and their lexical forms follow cfg_grammar/Namespace.g4; anything else
    (None, lists, exponents) is a ValueError, as it is a parse failure in the reference."""
    text = text.strip()
    head = "Namespace("
    if not text.startswith(head) or not text.endswith(")"):
        raise ValueError("cfg_args: expected Namespace(...)")
    pos, end = len(head), len(text) - 1
    out: Dict[str, object] = {}

    def token(p):
        m = _TOKEN.match(text, p, end)
        if m is None or m.end() == p:
            raise ValueError(f"cfg_args: cannot read a token at offset {p}: {text[p:p + 20]!r}")
        return m

    while True:
        m = token(pos)
        if m.lastgroup != "id":
            raise ValueError(f"cfg_args: expected a key at offset {pos}")
        key, pos = m.group("id"), m.end()
        m = token(pos)
        if m.group("punct") != "=":
            raise ValueError(f"cfg_args: expected '=' after {key}")
        m = token(m.end())
        kind, pos = m.lastgroup, m.end()
        if kind == "int":
            out[key] = int(m.group("int"))
        elif kind == "float":
            out[key] = float(m.group("float"))
        elif kind == "bool":
            out[key] = True if reference_bools else m.group("bool") in ("True", "true")      # parse_config.py:36: bool("False") is True
        elif kind == "str":
            out[key] = m.group("str")[1:-1]
        else:
            raise ValueError(f"cfg_args: value of {key} is not INT / FLOAT / BOOL / STRING")
        if text[pos:end].strip() == "":
            return out
        m = token(pos)
        if m.group("punct") != ",":
            raise ValueError(f"cfg_args: expected ',' after the value of {key}")
        pos = m.end()


def get_checkpoint_arguments(root_dir: str, reference_bools: bool = True) -> dotdict:
    with open(os.path.join(root_dir, "cfg_args")) as fh:
        return dotdict(parse_cfg_args(fh.read(), reference_bools))


# ---------------------------------------------------------------------------------------------------------------------------
# experiment discovery
# ---------------------------------------------------------------------------------------------------------------------------
def get_highest_valid_checkpoint(root_dir: str) -> str:
    """`<root>/point_cloud/iteration_<N>/point_cloud.ply` with the largest N; "" when there is none."""
    ckpt_dir = os.path.join(root_dir, "point_cloud")
    best, best_path = -1, ""
    for name in (os.listdir(ckpt_dir) if os.path.isdir(ckpt_dir) else ()):
        parts = name.split("_")
        if len(parts) < 2 or parts[0] != "iteration" or not re.fullmatch(r"[+-]?\d+", parts[1].strip()):
            continue
        path = os.path.join(ckpt_dir, name, "point_cloud.ply")
        if os.path.exists(path) and int(parts[1]) >= best:
            best, best_path = int(parts[1]), path
    return best_path


def parse_exp_dir(exp_dir: str, prefix: str) -> Dict[str, dict]:
    """Directories `<prefix>..._<sequence id>` of `exp_dir` that hold a checkpoint, keyed by sequence id, in name order."""
    found: Dict[str, dict] = {}
    for name in sorted(os.listdir(exp_dir)):
        path = os.path.join(exp_dir, name)
        if not (os.path.isdir(path) and name.startswith(prefix)):
            continue
        parts = name.split("_")
        sequence_id, category = parts[-1], "_".join(parts[:-1])
        ckpt = get_highest_valid_checkpoint(path)
        if ckpt == "":
            print(f"Object {sequence_id} of category {category} skipped because no valid checkpoint found")
            continue
        found[sequence_id] = {"exp_dir_filepath": path, "checkpoint_filepath": ckpt, "sequence_id": sequence_id, "category_name": category}
    return found


# ---------------------------------------------------------------------------------------------------------------------------
# camera helpers
# ---------------------------------------------------------------------------------------------------------------------------
def focal2fov(focal, pixels):      # utils/graphics_utils.py:83-84
    return 2 * math.atan(pixels / (2 * focal))


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


def qvec2rotmat(q):
    """COLMAP quaternion (w, x, y, z) -> rotation matrix."""
    w, x, y, z = q
    return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                     [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                     [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])


def nerfpp_norm(cams) -> dict:
    """scene/datasets_utils.py:18-31: centre of the camera centres and 1.1 x the largest distance from it."""
    if not cams:
        return {"translate": np.zeros(3), "radius": 0.0}
    centres = []
    for c in cams:
        w2c = np.eye(4)
        w2c[:3, :3] = np.asarray(c.R).T
        w2c[:3, 3] = np.asarray(c.T)
        # the reference goes world-to-view -> inverse -> inverse again and hands the matrix on as float32 (graphics_utils.py:42-53);
        # the camera centre is then read from the inverse of that single-precision matrix
        w2c32 = np.float32(np.linalg.inv(np.linalg.inv(w2c)))
        centres.append(np.linalg.inv(w2c32)[:3, 3])
    centres = np.stack(centres, axis=1)
    centre = centres.mean(axis=1, keepdims=True)
    radius = float(np.linalg.norm(centres - centre, axis=0).max()) * 1.1
    return {"translate": -centre.flatten(), "radius": radius}


def _open_image(path):
    from PIL import Image
    return Image.open(path)


def _composite_rgba(image, white_background: bool):
    """RGBA -> RGB on white / black as the T&T and Blender readers do (tanksandtemples.py:57-63): float composite, then the
    reference's `np.array(x * 255.0, dtype=np.byte)` -- truncation toward zero, kept as the byte pattern."""
    from PIL import Image
    data = np.array(image.convert("RGBA")) / 255.0
    bg = np.array([1.0, 1.0, 1.0]) if white_background else np.array([0.0, 0.0, 0.0])
    arr = data[:, :, :3] * data[:, :, 3:4] + bg * (1 - data[:, :, 3:4])
    return Image.fromarray((arr * 255.0).astype(np.int64).astype(np.uint8), "RGB")


# ---------------------------------------------------------------------------------------------------------------------------
# COLMAP (Mip-NeRF360 layout: <path>/sparse/0/{cameras,images}.{bin,txt}, <path>/images/)
# ---------------------------------------------------------------------------------------------------------------------------
_COLMAP_MODELS = {0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5), 4: ("OPENCV", 8), 5: ("OPENCV_FISHEYE", 8),
                  6: ("FULL_OPENCV", 12), 7: ("FOV", 5), 8: ("SIMPLE_RADIAL_FISHEYE", 4), 9: ("RADIAL_FISHEYE", 5), 10: ("THIN_PRISM_FISHEYE", 12)}


def read_colmap_cameras_binary(path):
    """cameras.bin: uint64 count; per camera int32 id, int32 model, uint64 width, uint64 height, float64 params[model]."""
    cams = {}
    with open(path, "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        for _ in range(n):
            cid, model, w, h = struct.unpack("<iiQQ", f.read(24))
            name, npar = _COLMAP_MODELS[model]
            cams[cid] = dict(id=cid, model=name, width=int(w), height=int(h), params=np.array(struct.unpack("<" + "d" * npar, f.read(8 * npar))))
    return cams


def read_colmap_images_binary(path):
    """images.bin: uint64 count; per image int32 id, float64 qvec[4], tvec[3], int32 camera id, NUL-terminated name, uint64 number of 2D
    points followed by (float64 x, float64 y, int64 point id) triples, which the pose path does not need and skips."""
    imgs = {}
    with open(path, "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        for _ in range(n):
            vals = struct.unpack("<idddddddi", f.read(64))
            name = bytearray()
            while True:
                ch = f.read(1)
                if ch in (b"\x00", b""):
                    break
                name += ch
            (npts,) = struct.unpack("<Q", f.read(8))
            f.seek(24 * npts, os.SEEK_CUR)
            imgs[vals[0]] = dict(id=vals[0], qvec=np.array(vals[1:5]), tvec=np.array(vals[5:8]), camera_id=vals[8], name=name.decode("utf-8"))
    return imgs


def read_colmap_cameras_text(path):
    cams = {}
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            e = line.split()
            if e[1] != "PINHOLE":       # colmap_utils.py:171: the text reader accepts PINHOLE only
                raise AssertionError("While the loader support other types, the rest of the code assumes PINHOLE")
            cams[int(e[0])] = dict(id=int(e[0]), model=e[1], width=int(e[2]), height=int(e[3]), params=np.array([float(v) for v in e[4:]]))
    return cams


def read_colmap_images_text(path):
    imgs = {}
    with open(path) as f:
        while True:
            line = f.readline()
            if not line:
                break
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            e = line.split()
            imgs[int(e[0])] = dict(id=int(e[0]), qvec=np.array([float(v) for v in e[1:5]]), tvec=np.array([float(v) for v in e[5:8]]),
                                   camera_id=int(e[8]), name=e[9])
            f.readline()                 # the 2D points of the image
    return imgs


def read_colmap_scene_info(path, images, eval, llffhold: int = 8) -> SceneInfo:
    """scene/colmap.py:73-131: binary model first, text model otherwise; cameras sorted by image name; with `eval` every
    llffhold-th camera is a test view."""
    sparse = os.path.join(path, "sparse/0")
    try:
        extr, intr = read_colmap_images_binary(os.path.join(sparse, "images.bin")), read_colmap_cameras_binary(os.path.join(sparse, "cameras.bin"))
    except Exception:
        extr, intr = read_colmap_images_text(os.path.join(sparse, "images.txt")), read_colmap_cameras_text(os.path.join(sparse, "cameras.txt"))
    folder = os.path.join(path, "images" if images is None else images)
    cams = []
    for e in extr.values():
        c = intr[e["camera_id"]]
        if c["model"] == "SIMPLE_PINHOLE":
            fovy, fovx = focal2fov(c["params"][0], c["height"]), focal2fov(c["params"][0], c["width"])
        elif c["model"] == "PINHOLE":
            fovy, fovx = focal2fov(c["params"][1], c["height"]), focal2fov(c["params"][0], c["width"])
        else:
            raise AssertionError("Colmap camera model not handled: only undistorted datasets (PINHOLE or SIMPLE_PINHOLE cameras) supported!")
        image_path = os.path.join(folder, os.path.basename(e["name"]))
        cams.append(CameraInfo(uid=c["id"], R=np.transpose(qvec2rotmat(e["qvec"])), T=np.array(e["tvec"]), FovY=fovy, FovX=fovx,
                               image=_open_image(image_path).convert("RGB"), image_path=image_path,
                               image_name=os.path.basename(image_path).split(".")[0], width=c["width"], height=c["height"]))
    cams.sort(key=lambda c: c.image_name)
    train = [c for i, c in enumerate(cams) if not eval or i % llffhold != 0]
    test = [c for i, c in enumerate(cams) if eval and i % llffhold == 0]
    return SceneInfo(None, train, test, nerfpp_norm(train), os.path.join(sparse, "points3D.ply"))


# ---------------------------------------------------------------------------------------------------------------------------
# NSVF / Tanks&Temples layout: intrinsics.txt, pose/<split>_*.txt (camera-to-world), rgb/<split>_*.png; split 0 train, 2 (else 1) test
# ---------------------------------------------------------------------------------------------------------------------------
def read_tanksandtemples_scene_info(path, eval, white_background=True, extension=".png", downsample=1.0) -> SceneInfo:
    """scene/tanksandtemples.py:110-167.  NOTE the reference's dispatcher calls this as (source_path, white_background, eval)
    (dataset_loader.py:18): the argument named `eval` receives white_background and vice versa; `eval` is unused either way, so
    only the background colour is affected -- load_data below passes the arguments in the reference's order."""
    intr = np.loadtxt(os.path.join(path, "intrinsics.txt"))[:3, :3]
    pose_files, img_files = sorted(os.listdir(os.path.join(path, "pose"))), sorted(os.listdir(os.path.join(path, "rgb")))

    def split(tag):
        return [x for x in pose_files if x.startswith(tag)], [x for x in img_files if x.startswith(tag)]

    def cameras(poses, imgs):
        out = []
        if not imgs:
            return out
        for idx, (img_name, pose_name) in enumerate(zip(imgs, poses)):
            w2c = np.linalg.inv(np.loadtxt(os.path.join(path, "pose", pose_name)))
            image_path = os.path.join(path, "rgb", img_name)
            image = _composite_rgba(_open_image(image_path), white_background)
            out.append(CameraInfo(uid=idx, R=np.transpose(w2c[:3, :3]), T=w2c[:3, 3], FovY=focal2fov(intr[1, 1], image.size[1]),
                                  FovX=focal2fov(intr[0, 0], image.size[0]), image=image, image_path=image_path, image_name=img_name,
                                  width=image.size[0], height=image.size[1]))
        return out

    train = cameras(*split("0_"))
    tp, ti = split("2_")
    if not tp:
        tp, ti = split("1_")
    test = cameras(tp, ti)
    return SceneInfo(None, train, test, nerfpp_norm(train), os.path.join(path, "points3d.ply"))


# ---------------------------------------------------------------------------------------------------------------------------
# Blender / NeRF-synthetic layout: transforms_{train,test}.json
# ---------------------------------------------------------------------------------------------------------------------------
def read_nerf_synthetic_info(path, white_background, eval, extension=".png") -> SceneInfo:
    """scene/synthetic.py:74-114: OpenGL camera-to-world matrices (Y up, Z back) -> COLMAP axes, RGBA composited on the background."""
    def cameras(name):
        out = []
        with open(os.path.join(path, name)) as fh:
            meta = json.load(fh)
        fovx = meta["camera_angle_x"]
        for idx, frame in enumerate(meta["frames"]):
            image_path = os.path.join(path, frame["file_path"] + extension)
            c2w = np.array(frame["transform_matrix"], dtype=np.float64)
            c2w[:3, 1:3] *= -1
            w2c = np.linalg.inv(c2w)
            image = _composite_rgba(_open_image(image_path), white_background)
            out.append(CameraInfo(uid=idx, R=np.transpose(w2c[:3, :3]), T=w2c[:3, 3], FovY=focal2fov(fov2focal(fovx, image.size[0]), image.size[1]),
                                  FovX=fovx, image=image, image_path=image_path, image_name=os.path.splitext(os.path.basename(image_path))[0],
                                  width=image.size[0], height=image.size[1]))
        return out

    train, test = cameras("transforms_train.json"), cameras("transforms_test.json")
    if not eval:
        train, test = train + test, []
    return SceneInfo(None, train, test, nerfpp_norm(train), os.path.join(path, "points3d.ply"))


# ---------------------------------------------------------------------------------------------------------------------------
# dispatch (scene/dataset_loader.py)
# ---------------------------------------------------------------------------------------------------------------------------
def load_data(args) -> SceneInfo:
    """`args`: the scene's training arguments (get_checkpoint_arguments): source_path, images, eval, white_background."""
    src = args.source_path
    if os.path.exists(os.path.join(src, "sparse")):
        return read_colmap_scene_info(src, args.images, args.eval)
    if os.path.exists(os.path.join(src, "transforms_train.json")):
        print("Found transforms_train.json file, assuming Blender data set!")
        return read_nerf_synthetic_info(src, args.white_background, args.eval)
    if os.path.exists(os.path.join(src, "intrinsics.txt")):
        print("Found intrinsics.txt file, assuming Tanks And Temple data set!")
        return read_tanksandtemples_scene_info(src, args.white_background, args.eval)      # positional, as dataset_loader.py:18
    raise AssertionError("Could not recognize scene type!")


def get_dataset_prefix(source_path: str) -> str:
    for marker, prefix in (("sparse", "mip_360"), ("transforms_train.json", "synthetic"), ("intrinsics.txt", "tt"), ("reconstruction.nvm", "cl")):
        if os.path.exists(os.path.join(source_path, marker)):
            return prefix
    raise AssertionError("Could not recognize scene type!")
