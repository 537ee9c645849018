"""train_id_module -- the on-box training of the scorer (pose_estimation/train.py:16-317), SURVEY 8(f)#1.

No pretrained id_module.th ships with the reference: every scene trains its own IdentificationModule for 1500 iterations of 32
accumulated single-image steps on rays re-emitted every 10 iterations.  The loop below follows the reference step for step
(same sampling calls in the same order, same loss terms and weights, same optimiser, same checkpoint layout) on top of this
build's pieces: rays from the HIP emitter (`rays_generator` is `functools.partial(generate_all_possible_rays, model)` as in
pretrain_eval_attention.py:73), the differentiable `IdentificationModule.forward` (PyTorch-ROCm autograd), the HIP target
scores of `DistanceBasedScoreLoss`, and the HIP inference path for the periodic evaluation (`test_pose_estimation`).
TensorBoard logging is optional (the package is not a dependency): without it the scalars go to `log_fn` / nowhere.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np
import torch

from .distance_based_loss import DistanceBasedScoreLoss
from .test import gt_pose_and_intrinsics, test_pose_estimation


class _NoWriter:
    def add_text(self, *a, **k):
        pass

    def add_scalar(self, *a, **k):
        pass


def _writer():
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter()
    except Exception:
        return _NoWriter()


def prepare_training_image(image, device):
    """train.py:109-121: uint8 [H,W,3|4] -> fp32 image in [0,1] (RGBA composited on white) and the alpha > 0.3 mask."""
    img = torch.from_numpy(np.array(image)).to(device=device, dtype=torch.float32) / 255.0
    if img.shape[-1] == 4:
        mask = img[..., -1] > 0.3
        img = torch.multiply(img[..., :3], img[..., -1:]) + (1 - img[..., -1:])
    else:
        mask = torch.ones_like(img[..., -1], dtype=torch.bool, device=device)
    return img, mask


def training_step_loss(id_module, loss_fn, camera_info, rays_ori, rays_dirs, rays_rgb, model_up, device):
    """One accumulated step (train.py:106-170): forward on one training image, score loss + 0.1 x camera-up loss.
    Returns (combined loss, score loss, camera-up loss) as tensors (combined may be NaN: the caller skips it)."""
    img, mask = prepare_training_image(camera_info.image, device)
    c2w, K = gt_pose_and_intrinsics(camera_info, device)
    scores, attn_map, _, up, rays_idx = id_module(img, mask, rays_ori, rays_dirs, rays_rgb)
    loss_score, _ = loss_fn(scores, c2w.to(device), K.to(device), rays_ori[rays_idx], rays_dirs[rays_idx], attn_map.shape[-2],
                            id_module.backbone_wrapper.backbone_wh, model_up=model_up)
    cam_up = -0.5 * torch.cosine_similarity(model_up, up, dim=-1) + 0.5
    return loss_score + 0.1 * cam_up, loss_score, cam_up


def train_id_module(ckpt_path, device, id_module, rays_generator: Optional[Callable[[], Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]],
                    scene_info, sequence_id, category_id, start_iterations: int = 0, renewal_every_n_iterations: int = 10,
                    display_every_n_iterations: int = 20, val_every_n_iterations: int = 20, n_iterations: int = 1500,
                    gradient_accumulation_steps: int = 32, lock_backbone: bool = True, log_fn: Optional[Callable[[str, float, int], None]] = None):
    from transformers.optimization import Adafactor        # the reference's optimiser (train.py:13,42-47), default arguments

    id_module.train()
    extra = []
    if lock_backbone:
        id_module.backbone_wrapper.eval()
    else:
        extra = list(id_module.backbone_wrapper.parameters())
    optimizer = Adafactor(list(id_module.ray_preprocessor.parameters()) + list(id_module.attention.parameters())
                          + list(id_module.camera_direction_prediction_network.parameters()) + extra)
    loss_fn = DistanceBasedScoreLoss()
    writer = _writer()
    for k, v in (("ckpt_path", ckpt_path), ("category_id", category_id), ("sequence_id", sequence_id)):
        writer.add_text("config/" + k, str(v))

    def log(tag, value, step):
        writer.add_scalar(tag, value, global_step=step)
        if log_fn is not None:
            log_fn(tag, float(value), step)

    model_up = torch.from_numpy(np.mean(np.asarray([c.R[:3, 1] for c in scene_info.train_cameras], dtype=np.float32), axis=0)).to(device)
    rays_ori = rays_dirs = rays_rgb = None
    running_loss = 0.0
    for iteration in range(start_iterations, n_iterations):
        if iteration % renewal_every_n_iterations == 0:
            rays_ori, rays_dirs, rays_rgb = rays_generator()
        optimizer.zero_grad()
        img_idx = torch.randint(0, len(scene_info.train_cameras), (gradient_accumulation_steps,), dtype=torch.long, device=device)
        acc_loss = acc_up = acc_score = 0.0
        for step in range(gradient_accumulation_steps):
            cam = scene_info.train_cameras[img_idx[step]]
            combined, loss_score, cam_up = training_step_loss(id_module, loss_fn, cam, rays_ori, rays_dirs, rays_rgb, model_up, device)
            if combined.isnan().any():
                continue
            (combined / gradient_accumulation_steps).backward()
            acc_loss += combined.item()
            acc_up += cam_up.item() / gradient_accumulation_steps
            acc_score += loss_score.item() / gradient_accumulation_steps
        optimizer.step()
        id_module.invalidate_caches()                 # the inference-side packed weights / key cache follow the parameters
        log("train/loss", acc_loss, iteration)
        log("train/cam_up", acc_up, iteration)
        log("train/loss_score", acc_score, iteration)
        running_loss += acc_loss
        if iteration % display_every_n_iterations == display_every_n_iterations - 1:
            print(f"[{iteration}] loss: {running_loss / display_every_n_iterations}")
            running_loss = 0.0
        if iteration % val_every_n_iterations == val_every_n_iterations - 1:
            for split, cams in (("train", scene_info.train_cameras), ("val", scene_info.test_cameras)):
                print(f"Eval on {'validation' if split == 'val' else split}...")
                _, te, ae, sc, rc = test_pose_estimation(cams, id_module, rays_ori, rays_dirs, rays_rgb, model_up, sequence_id=sequence_id,
                                                         category_id=category_id, loss_fn=loss_fn)
                for tag, v in (("avg_translation_error", te), ("avg_angular_error", ae), ("avg_loss_score", sc), ("recall", rc)):
                    log(f"{split}/{tag}", v, iteration)
            id_module.train()
            if lock_backbone:
                id_module.backbone_wrapper.eval()
    torch.save({"epoch": n_iterations, "model_state_dict": id_module.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
                "running_loss": running_loss}, ckpt_path)
