"""Scene container with the getters the pose path uses from the reference's GaussianModel
(scene/gaussian_model.py:125-163), plus the CameraInfo tuple of scene/scene_structure.py:7-17.

Only the raw parameter tensors live here (SoA: one contiguous fp32 array per attribute, 236 B per
Gaussian); the activations (exp / normalize / quaternion->R) are fused into the HIP kernels, so
`get_scaling`, `get_rotation_mat` ... exist for API parity but the emitter never materialises them.
"""
from __future__ import annotations

from typing import NamedTuple, Optional

import os

import numpy as np
import torch


class CameraInfo(NamedTuple):
    uid: int
    R: np.ndarray
    T: np.ndarray
    FovY: float
    FovX: float
    image: object
    image_path: str
    image_name: str
    width: int
    height: int


class GaussianScene:
    """Drop-in for the attributes of GaussianModel read by generate_all_possible_rays
    (sampling.py:133-238): get_xyz, get_scaling, get_rotation_mat, get_features, active_sh_degree,
    max_sh_degree."""

    def __init__(self, sh_degree: int = 3):
        self.max_sh_degree = sh_degree
        self.active_sh_degree = sh_degree
        e = torch.empty(0)
        self._xyz = e
        self._scaling = e
        self._rotation = e
        self._features_dc = e
        self._features_rest = e
        self._opacity = e

    # ---- construction ---------------------------------------------------------------------------
    @classmethod
    def from_arrays(cls, xyz, log_scale, rot, f_dc, f_rest, opacity=None, sh_degree: Optional[int] = None, device="cuda"):
        def t(a):
            a = torch.as_tensor(np.ascontiguousarray(a) if isinstance(a, np.ndarray) else a, dtype=torch.float32)
            return a.to(device).contiguous()

        n_coef = 1 + f_rest.shape[1]
        deg = int(round(n_coef ** 0.5)) - 1 if sh_degree is None else int(sh_degree)
        s = cls(deg)
        s._xyz, s._scaling, s._rotation = t(xyz), t(log_scale), t(rot)
        s._features_dc, s._features_rest = t(f_dc), t(f_rest)
        s._opacity = t(opacity) if opacity is not None else torch.zeros(s._xyz.shape[0], 1, device=device)
        return s

    @classmethod
    def from_dict(cls, d, device="cuda"):
        return cls.from_arrays(d["xyz"], d["log_scale"], d["rot"], d["f_dc"], d["f_rest"], d.get("opacity"),
                               int(d["sh_degree"]) if "sh_degree" in d else None, device)

    # ---- 3DGS PLY (gaussian_model.py:284-340 writes it, :342-420 reads it) --------------------------
    # One "vertex" element of float32 properties x y z nx ny nz f_dc_0..2 f_rest_0..(3*((deg+1)^2-1)-1) opacity scale_0..2
    # rot_0..3, binary little endian.  f_dc / f_rest are stored channel-major ([N,3,coeffs] flattened) and transposed to
    # the [N,coeffs,3] layout of the model on load; scales are logs, rotations un-normalised (w,x,y,z), opacity a logit.
    @staticmethod
    def _ply_properties(n_rest: int):
        names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(n_rest)]
        return names + ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]

    @classmethod
    def load_ply(cls, path: str, sh_degree: int = 3, device="cuda"):
        """Reads a 3DGS point_cloud.ply straight into the device arrays (no plyfile dependency; any property order,
        float32 or float64 properties, little- or big-endian binary)."""
        types = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "uchar": "u1", "uint8": "u1", "char": "i1", "int8": "i1",
                 "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4"}
        with open(path, "rb") as f:
            if f.readline().strip() != b"ply":
                raise RuntimeError(f"{path}: not a PLY file")
            fmt, n, props, in_vertex = None, None, [], False
            while True:
                line = f.readline()
                if not line:
                    raise RuntimeError(f"{path}: truncated PLY header")
                tok = line.decode("ascii", "replace").split()
                if not tok or tok[0] == "comment":
                    continue
                if tok[0] == "format":
                    fmt = tok[1]
                elif tok[0] == "element":
                    if n is not None and tok[1] != "vertex":
                        in_vertex = False
                        continue
                    in_vertex = tok[1] == "vertex"
                    if in_vertex:
                        n = int(tok[2])
                elif tok[0] == "property" and in_vertex:
                    if tok[1] == "list":
                        raise RuntimeError(f"{path}: list property in the vertex element")
                    props.append((tok[2], types[tok[1]]))
                elif tok[0] == "end_header":
                    break
            if fmt not in ("binary_little_endian", "binary_big_endian") or n is None:
                raise RuntimeError(f"{path}: need a binary PLY with a vertex element (format {fmt})")
            end = "<" if fmt == "binary_little_endian" else ">"
            data = np.fromfile(f, dtype=np.dtype([(nm, end + t) for nm, t in props]), count=n)
        if data.shape[0] != n:
            raise RuntimeError(f"{path}: {data.shape[0]} of {n} vertices present")
        names = set(data.dtype.names)

        def cols(prefix):
            c = sorted((nm for nm in names if nm.startswith(prefix)), key=lambda x: int(x.split("_")[-1]))
            return np.stack([data[nm].astype(np.float32) for nm in c], axis=1) if c else np.zeros((n, 0), np.float32)

        n_rest = 3 * (sh_degree + 1) ** 2 - 3
        rest = cols("f_rest_")
        if rest.shape[1] != n_rest:                                    # gaussian_model.py:368
            raise RuntimeError(f"{path}: {rest.shape[1]} f_rest properties, SH degree {sh_degree} needs {n_rest}")
        xyz = np.stack([data[a].astype(np.float32) for a in ("x", "y", "z")], axis=1)
        f_dc = cols("f_dc_").reshape(n, 3, 1).transpose(0, 2, 1)
        f_rest = rest.reshape(n, 3, (sh_degree + 1) ** 2 - 1).transpose(0, 2, 1)
        s = cls.from_arrays(xyz, cols("scale_"), cols("rot"), f_dc, f_rest, data["opacity"].astype(np.float32)[:, None], sh_degree, device)
        s.active_sh_degree = s.max_sh_degree                            # gaussian_model.py:420
        return s

    def save_ply(self, path: str):
        """Writes the layout of gaussian_model.py:298-334 (normals are zeros there too)."""
        n = len(self)
        f_dc = self._features_dc.detach().transpose(1, 2).flatten(start_dim=1).cpu().numpy()
        f_rest = self._features_rest.detach().transpose(1, 2).flatten(start_dim=1).cpu().numpy()
        names = self._ply_properties(f_rest.shape[1])
        attrs = np.concatenate([self._xyz.detach().cpu().numpy(), np.zeros((n, 3), np.float32), f_dc, f_rest,
                                self._opacity.detach().cpu().numpy().reshape(n, 1), self._scaling.detach().cpu().numpy(),
                                self._rotation.detach().cpu().numpy()], axis=1).astype("<f4")
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "wb") as f:
            f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n).encode())
            f.write("".join(f"property float {nm}\n" for nm in names).encode())
            f.write(b"end_header\n")
            f.write(np.ascontiguousarray(attrs).tobytes())

    def to(self, device):
        for k in ("_xyz", "_scaling", "_rotation", "_features_dc", "_features_rest", "_opacity"):
            setattr(self, k, getattr(self, k).to(device))
        return self

    @property
    def device(self):
        return self._xyz.device

    def __len__(self):
        return self._xyz.shape[0]

    # ---- getters of the reference (gaussian_model.py:125-163) -------------------------------------
    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    def nbytes(self) -> int:
        return sum(getattr(self, k).numel() * 4 for k in ("_xyz", "_scaling", "_rotation", "_features_dc", "_features_rest",
                                                          "_opacity"))


GaussianModel = GaussianScene
