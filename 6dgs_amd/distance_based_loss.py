"""DistanceBasedScoreLoss (pose_estimation/distance_based_loss.py:147-283) on the HIP path.

Same constructor and `forward(pred_score, camera_pose, camera_intrinsic, rays_ori, rays_dir, total_number_of_features,
backbone_wh, model_up=None, obs_img_shape=(800, 800)) -> (avg_score, combined_score)` as the reference, so it can be handed to
`test_pose_estimation(loss_fn=...)` (evaluation: `pretrain_eval_attention.py`) or to the reference's training loop.  The target
scores -- one pass over all rays -- come from `sixdgs_distance_target`; the loss value `mean((pred - target)^2)` is formed with
PyTorch ops so that it stays differentiable with respect to `pred_score`.  The reference also projects the ray origins into the
feature grid (`is_inside`, distance_based_loss.py:73-121) but never uses the result; that part is not computed.
"""
from __future__ import annotations

from typing import Tuple

import torch

from . import ops


class DistanceBasedScoreLoss(torch.nn.Module):
    def __init__(self, reweight_method="none", lds=False, lds_kernel="gaussian", lds_ks=5, lds_sigma=2,
                 total_number_of_elements: float = 256.0):
        super().__init__()
        assert reweight_method in {"none", "inverse", "sqrt_inv"}
        assert reweight_method != "none" if lds else True, "Set reweight to 'sqrt_inv' (default) or 'inverse' when using LDS"
        self.reweight_method, self.lds, self.lds_kernel, self.lds_ks, self.lds_sigma = reweight_method, lds, lds_kernel, lds_ks, lds_sigma

    def forward(self, pred_score: torch.Tensor, camera_pose: torch.Tensor, camera_intrinsic: torch.Tensor, rays_ori: torch.Tensor,
                rays_dir: torch.Tensor, total_number_of_features: int, backbone_wh: Tuple[int, int], model_up=None,
                obs_img_shape=(800, 800)):
        with torch.no_grad():
            combined_score = ops.distance_target(rays_ori, rays_dir, camera_pose, int(total_number_of_features))
        avg_score = torch.square(pred_score - combined_score).mean()
        return avg_score, combined_score
