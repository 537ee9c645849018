"""TEST INFRASTRUCTURE ONLY.  A deterministic stand-in for the `loss_fn` argument of test_pose_estimation
(pose_estimation/test.py:108-142 calls it as loss_fn(pred_scores, pose, K, rays_ori, rays_dirs, n_tokens, backbone_wh,
model_up=...) -> (scalar, target_scores[R])).  The reference's own DistanceBasedScoreLoss needs the training stack; this
callable has the same signature and the same kind of output (rays that pass close to the ground-truth camera centre, in
front of it, score high), so oracle/gen_golden.py can run the REFERENCE loop with it and tests/ can run the build's loop with
the identical callable: what is pinned is the control flow of that branch (argument order, recall quirk, substitution of the
ground-truth top-k for the prediction)."""
import torch


def line_distance_loss(pred_scores, pose, K, rays_ori, rays_dirs, n_tokens, backbone_wh, model_up=None):
    assert K.shape == (3, 3) and int(n_tokens) > 0 and int(backbone_wh[0]) > 0
    c = pose[:3, 3]
    v = c[None, :] - rays_ori
    t = (v * rays_dirs).sum(dim=-1, keepdim=True)
    d = torch.linalg.norm(v - t * rays_dirs, dim=-1)
    target = torch.exp(-4.0 * d) * (t[:, 0] > 0).to(d.dtype)
    loss = ((pred_scores / float(n_tokens) - target / target.sum()) ** 2).sum()
    return loss, target
