"""Scene container with the getters the pose path uses from the reference's GaussianModel
(scene/gaussian_model.py:125-163), plus the CameraInfo tuple of scene/scene_structure.py:7-17.

Only the raw parameter tensors live here (SoA: one contiguous fp32 array per attribute, 236 B per
Gaussian); the activations (exp / normalize / quaternion->R) are fused into the HIP kernels, so
`get_scaling`, `get_rotation_mat` ... exist for API parity but the emitter never materialises them.
"""
from __future__ import annotations

from typing import NamedTuple, Optional

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
