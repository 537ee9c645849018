"""test_pose_estimation -- drop-in for pose_estimation/test.py:23-323 (the per-image evaluation loop).

Same signature, same results-dict schema (test.py:290-302), same return tuple, same printed summary.
The loop body is restated batched: all query images of a batch go through the backbone together, are
scored against ONE pass over the cached ray keys, and the filter / line-intersection / rotation
assembly / error metrics run for the whole batch in one kernel launch -- the ~15 device syncs per
image of the reference (`.item()`, boolean indexing, `unique`, CPU eye(3) fills) become one D2H copy
per batch.
"""
from __future__ import annotations

import math
import os
import time
from statistics import mean
from typing import List, Optional

import numpy as np
import torch

from . import ops

_U8_TO_F32 = None


def _u8_lut(device):
    """uint8 -> fp32 / 255.0 with the CPU reference's true division (torch's GPU `x / 255.0` multiplies
    by the reciprocal and differs in the last ulp for some values): a 256-entry table built on the CPU."""
    global _U8_TO_F32
    if _U8_TO_F32 is None or _U8_TO_F32.device != torch.device(device):
        _U8_TO_F32 = (torch.arange(256, dtype=torch.float32) / 255.0).to(device)
    return _U8_TO_F32


def fov2focal(fov, pixels):  # utils/graphics_utils.py:79-80
    return pixels / (2 * math.tan(fov / 2))


def prepare_image(image, device):
    """test.py:69-83: uint8 HxWx{3,4} -> (img fp32 [H,W,3] in [0,1], mask bool [H,W])."""
    arr = torch.from_numpy(np.ascontiguousarray(np.array(image)))
    if arr.dtype != torch.uint8:
        obs = arr.to(device=device, dtype=torch.float32) / 255.0
    else:
        obs = _u8_lut(device)[arr.to(device).long()]
    if obs.shape[-1] == 4:
        mask = obs[..., -1] > 0.3
        obs = torch.multiply(obs[..., :3], obs[..., -1:]) + (1 - obs[..., -1:])
    else:
        mask = torch.ones_like(obs[..., -1], dtype=torch.bool)
    return obs, mask


def prepare_image_device(img_u8: torch.Tensor):
    """Same as prepare_image for a uint8 tensor that already lives on the GPU."""
    obs = _u8_lut(img_u8.device)[img_u8.long()]
    if obs.shape[-1] == 4:
        mask = obs[..., -1] > 0.3
        obs = torch.multiply(obs[..., :3], obs[..., -1:]) + (1 - obs[..., -1:])
    else:
        mask = torch.ones_like(obs[..., -1], dtype=torch.bool)
    return obs, mask


def prepare_images_device(images):
    """uint8 device images (a list, or one stacked [B,H,W,C] tensor) -> (fp32 images, masks) with mask None for RGB (all-ones mask, test.py:80-83).
    RGB images of one shape are converted in ONE launch (ops.u8_to_planar: the 256-entry table of _u8_lut, written planar) and come back as one
    [B,H,W,3] tensor -- a channels-last VIEW of the planar buffer, which BackboneWrapper.preprocess_batch permutes straight back (no copy on the way
    to the resize); it indexes / iterates like the list of images it replaces."""
    n = len(images)
    if n >= 1 and all(im.shape == images[0].shape and im.shape[-1] == 3 and im.dtype == torch.uint8 and im.is_cuda for im in images) \
            and (images[0].shape[0] * images[0].shape[1]) % 4 == 0:
        stacked = images if isinstance(images, torch.Tensor) else (images[0][None] if n == 1 else torch.stack(list(images)))
        planar = ops.u8_to_planar(stacked, _u8_lut(stacked.device))
        return planar.permute(0, 2, 3, 1), [None] * n
    out_i, out_m = [], []
    for im in images:
        o, m = prepare_image_device(im)
        out_i.append(o)
        out_m.append(m if im.shape[-1] == 4 else None)
    return out_i, out_m


def image_side_tokens(id_module, images):
    """uint8 device images -> (tokens, feature maps): RGB images of one shape go from uint8 to normalised 224 x 224 planes in one kernel
    (IdentificationModule.image_tokens_u8); everything else -- alpha channels, mixed sizes, shapes that kernel does not take -- through
    prepare_images_device + image_tokens as before."""
    n = len(images)
    if n >= 1 and ops.image_prep_enabled() and all(im.shape == images[0].shape and im.dim() == 3 and im.shape[-1] == 3 and im.dtype == torch.uint8 and im.is_cuda
                                                 for im in images):
        stacked = images if isinstance(images, torch.Tensor) else (images[0][None] if n == 1 else torch.stack(list(images)))
        res = id_module.image_tokens_u8(stacked, _u8_lut(stacked.device))
        if res is not None:
            return res
    imgs_f, masks = prepare_images_device(images)
    return id_module.image_tokens(imgs_f, masks)


class _ImageSideGraph:
    """hipGraphs of the image side of a batch: (a) uint8 -> fp32, resize / crop / normalise, ViT-S/14, token assembly; (b) the camera-up CNN on
    (a)'s feature maps -- ~250 small launches that are launch-bound at 1..16 images.  One pair per (module weights, batch shape); inputs are
    copied into a static buffer, outputs are static tensors consumed in stream order before the next replay.  Two graphs because the tokens
    are needed at the START of the scorer and the camera-up vector only by the pose solve at its END: the pipeline (PoseStream) takes the CNN
    off the path to the sweep (`defer_cnn`)."""

    MAX_ENTRIES = 4         # batch shapes kept: an evaluation's balanced batches come in two sizes per scene (19 views: 10 + 9), a sweep in a few more

    def __init__(self):
        self.key, self.g_vit, self.g_cnn, self.inp, self.tokens, self.fmaps, self.up, self.failed = None, None, None, None, None, None, None, False
        self.entries = {}       # key -> (g_vit, g_cnn, inp, tokens, fmaps, up), most recently used last

    def _activate(self, key):
        self.g_vit, self.g_cnn, self.inp, self.tokens, self.fmaps, self.up = self.entries[key]
        self.entries[key] = self.entries.pop(key)
        self.key = key

    def run(self, id_module, images, defer_cnn: bool = False):
        """-> (tokens, up), or (tokens, None) with defer_cnn (then `cnn()` replays the second graph); None when the batch cannot be captured."""
        if self.failed or len(images) < 1 or not all(im.shape == images[0].shape and im.shape[-1] == 3 and im.dtype == torch.uint8 for im in images):
            return None
        key = (len(images), tuple(images[0].shape), str(images[0].device), next(id_module.parameters()).data_ptr(),
               tuple(p._version for p in id_module.parameters()))
        try:
            if self.key != key and key in self.entries:
                self._activate(key)
                torch.stack(list(images), out=self.inp)
            elif self.key != key:
                self.inp = torch.stack(list(images))
                cur = torch.cuda.current_stream()
                side = torch.cuda.Stream()
                side.wait_stream(cur)
                with torch.cuda.stream(side):               # warm-up on a side stream (lazy initialisation, allocator)
                    id_module.camera_up(self._vit(id_module)[1])
                cur.wait_stream(side)
                g1 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g1):
                    self.tokens, self.fmaps = self._vit(id_module)
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, pool=g1.pool()):
                    self.up = id_module.camera_up(self.fmaps)
                if isinstance(self.tokens, (list, tuple)):   # ragged tokens (cannot happen for RGB views of one size): not a static output
                    raise RuntimeError("ragged tokens")
                self.g_vit, self.g_cnn, self.key = g1, g2, key
                self.entries = {k: v for k, v in self.entries.items() if k[3:] == key[3:]}      # graphs of other weights are stale
                self.entries[key] = (g1, g2, self.inp, self.tokens, self.fmaps, self.up)
                while len(self.entries) > self.MAX_ENTRIES:
                    self.entries.pop(next(iter(self.entries)))
            else:
                torch.stack(list(images), out=self.inp)
            self.g_vit.replay()
            if defer_cnn:
                return self.tokens, None
            self.g_cnn.replay()
            return self.tokens, self.up
        except Exception:                                   # capture unsupported (e.g. a backbone with host syncs): stay eager
            self.failed, self.key, self.g_vit, self.g_cnn, self.entries = True, None, None, None, {}
            return None

    def cnn(self):
        """The camera-up CNN on the feature maps of the last `run(..., defer_cnn=True)`."""
        self.g_cnn.replay()
        return self.up

    def _vit(self, id_module):
        return image_side_tokens(id_module, self.inp)


@torch.no_grad()
def prime_image_graph(id_module, images) -> bool:
    """Builds (or refreshes) the cached hipGraph of the image side for this batch shape ahead of time -- set-up work, like
    the key cache; estimate_poses builds it lazily otherwise.  Returns whether a graph is in use."""
    cache = id_module.__dict__.setdefault("_image_side_graph", _ImageSideGraph())
    return cache.run(id_module, images) is not None


@torch.no_grad()
def estimate_poses(id_module, images, rays_ori, rays_dirs, rays_rgb, gt_c2w=None, k: int = 100, workspace=None,
                   images_in_flight=None, profile=None, tokens=None, up=None, want_scores: bool = False, image_graph: bool = True,
                   streamed_chunk_rays: Optional[int] = None, defer_status: bool = False):
    """One batch of the hot path: query images (uint8 [H,W,3|4] tensors on the GPU) -> poses.
    image prep -> backbone tokens + camera-up (PyTorch-ROCm) -> q_proj / scorer / top-k / pose solve (HIP).
    `tokens` / `up` inject the image-side boundary inputs instead.  Everything is enqueued on the current
    stream; nothing syncs until the caller reads the returned device tensors.  `streamed_chunk_rays` selects the
    streamed scorer (IdentificationModule.score_tokens_streamed) with that chunk size.

    defer_status: the select path's per-image status is not read inside (no sync in the middle of the batch: the whole call is
    one capturable stream of launches); the result then carries `packed` = [c2w (16) | select status] per image -- ONE D2H for
    the caller -- and `resolve_poses(id_module, sol, packed_host)` re-does the images the select path refused (rare)."""
    if tokens is None:
        res = None
        if image_graph and not torch.cuda.is_current_stream_capturing():
            cache = id_module.__dict__.setdefault("_image_side_graph", _ImageSideGraph())
            res = cache.run(id_module, images)
        if res is not None and not isinstance(res[0], (list, tuple)):
            tokens, up = res
        else:
            tokens, fmaps = image_side_tokens(id_module, images)
            up = id_module.camera_up(fmaps)
    if streamed_chunk_rays:       # no resident key cache: ray chunks through ray MLP + scorer (scenes beyond one GPU's HBM)
        idx, weights = id_module.score_tokens_streamed(tokens, rays_ori, rays_dirs, rays_rgb, k, chunk_rays=int(streamed_chunk_rays),
                                                       profile=profile)
        scores = None
    else:
        idx, weights, scores = id_module.score_tokens(tokens, rays_ori, rays_dirs, rays_rgb, k, want_scores=want_scores,
                                                      workspace=workspace, images_in_flight=images_in_flight, profile=profile,
                                                      defer_status=defer_status)
    return _solve_batch(id_module, idx, weights, scores, tokens, up, rays_ori, rays_dirs, gt_c2w, defer_status)


def _solve_batch(id_module, idx, weights, scores, tokens, up, rays_ori, rays_dirs, gt_c2w, defer_status):
    """The end of a batch: pose solve on the selected rays; with defer_status the select statuses ride along with the poses (`packed`)."""
    sol = ops.solve_pose(rays_ori, rays_dirs, idx, weights, up, gt_c2w)
    sol.update(idx=idx, weights=weights, scores=scores, tokens=tokens, up=up)
    if defer_status:
        pend = getattr(id_module, "pending_select", None)
        st = pend["status"].to(torch.float32) if pend is not None else torch.zeros(idx.shape[0], device=idx.device)
        sol["packed"] = torch.cat([sol["c2w"].reshape(-1, 16), st[:, None]], dim=1)
        sol["pending_select"] = pend
        sol["_inputs"] = (rays_ori, rays_dirs, gt_c2w)
    return sol


@torch.no_grad()
def resolve_poses(id_module, sol, packed_host):
    """After the one D2H of a defer_status batch: c2w [B,4,4] on the host; images whose select status is negative (the bounds could
    not decide them) are scored by the two-pass scorer and solved again -- eagerly, outside any graph."""
    c2w = packed_host[:, :16].reshape(-1, 4, 4).clone()
    pend = sol.get("pending_select")
    if pend is None:
        return c2w
    status = packed_host[:, 16].round().to(torch.int64).tolist()
    if min(status) >= 0:
        id_module.last_select_candidates = status
        return c2w
    idx, weights = sol["idx"].clone(), sol["weights"].clone()
    idx, weights, redo = id_module.finish_select(idx, weights, pend, status_host=status)
    rays_ori, rays_dirs, gt = sol["_inputs"]
    again = ops.solve_pose(rays_ori, rays_dirs, idx, weights, sol["up"], gt)
    sol.update(again, idx=idx, weights=weights)
    return again["c2w"].cpu()


class PoseStream:
    """Batches of query images through `estimate_poses` as a PIPELINE: the reference's evaluation is exactly such a stream of views
    (pose_estimation/test.py:46-302), and poses/s over a test set is what the metric counts.

      handle = ps.submit(images, gt)      # enqueue everything of this batch; returns at once
      c2w, sol = ps.collect(handle)       # later: the batch's poses on the host

    With one batch submitted before the previous one is collected, (a) the host never sits between the GPU's last kernel of batch N and its
    first of batch N + 1 (round 4: every step ended in a device sync before the next image side was even enqueued), and (b) the image side of
    batch N + 1 -- ~250 small kernels, ViT-S/14 + camera-up CNN -- runs on its OWN stream, so it fills the compute units the sweep of batch N
    frees in its tail and overlaps batch N's small serial kernels (finish of U, selections, re-score, pose solve, D2H).  The scorer itself
    stays in order on the caller's stream (one sweep holds every CU: its registers and LDS leave room for nothing else).
    Poses are bit-identical to the unpipelined run: the same kernels on the same data in the same order per batch.
    Round 6 built and measured three further arrangements -- the tail of a batch on a third stream, the sweep on a CU-masked stream with the image and
    tail streams on the CUs it leaves out, the image side one batch further ahead -- none faster than this one (profiles/r06_pipeline_ab.md; commit b2320e2).

    Inputs of batch N must exist before submit(N - 1) was called, or be produced on `ps.image_stream` (uploads under
    `with torch.cuda.stream(ps.image_stream)`): the image stream does not wait for the caller's stream beyond that point.
    The image-side hipGraphs and their static token / camera-up buffers are the MODULE's (one capture per batch shape serves every scene of a sweep): the first
    submit waits for the caller's stream, so a PoseStream may follow `estimate_poses` calls on the same module; do not call `estimate_poses(image_graph=True)`
    BETWEEN the submits of a PoseStream without a host sync in between -- its replay would rewrite buffers the other stream may still read."""

    def __init__(self, id_module, rays_ori, rays_dirs, rays_rgb, k: int = 100, workspace=None, images_in_flight=None):
        self.idm, self.rays, self.k = id_module, (rays_ori, rays_dirs, rays_rgb), k
        self.workspace, self.images_in_flight = workspace, images_in_flight
        self.image_stream = torch.cuda.Stream(device=rays_ori.device)      # (a high-priority image stream was measured in round 6: no difference, profiles/r06_pipeline_ab.md)
        self._fence = None            # recorded on the caller's stream at the START of the previous submit

    @torch.no_grad()
    def submit(self, images, gt_c2w=None, profile=None, tokens=None, up=None):
        main = torch.cuda.current_stream()
        side = self.image_stream
        if tokens is None:
            if self._fence is None:
                side.wait_stream(main)
            else:
                side.wait_event(self._fence)
            fence = torch.cuda.Event()
            fence.record(main)
            self._fence = fence
            with torch.cuda.stream(side):
                cache = self.idm.__dict__.setdefault("_image_side_graph", _ImageSideGraph())
                res = cache.run(self.idm, images, defer_cnn=True)
                if res is not None:
                    # the graphs' outputs are STATIC buffers, rewritten by the next replay: this batch keeps its own copies (1.6 MB at 4 images).
                    # The tokens are ready when the ViT graph is; the camera-up CNN (needed only by the pose solve at the END of the batch) is replayed
                    # behind them and no longer sits on the path to the sweep
                    tk = res[0]
                    tokens = type(tk)(tk.feats.clone(), tk.pe) if hasattr(tk, "feats") else tk.clone()
                    ready = torch.cuda.Event()
                    ready.record(side)
                    up = cache.cnn().clone()
                    up_ready = torch.cuda.Event()
                    up_ready.record(side)
                else:
                    tokens, fmaps = image_side_tokens(self.idm, images)
                    up = self.idm.camera_up(fmaps)
                    ready = torch.cuda.Event()
                    ready.record(side)
                    up_ready = ready
            main.wait_event(ready)
            for t in ([tokens.feats] if hasattr(tokens, "feats") else ([tokens] if torch.is_tensor(tokens) else list(tokens))) + [up]:
                t.record_stream(main)          # allocated on the image stream, read on the caller's
        else:
            up_ready = None
        idx, weights, scores = self.idm.score_tokens(tokens, *self.rays, self.k, want_scores=False, workspace=self.workspace,
                                                     images_in_flight=self.images_in_flight, profile=profile, defer_status=True)
        if up_ready is not None:
            main.wait_event(up_ready)
        sol = _solve_batch(self.idm, idx, weights, scores, tokens, up, self.rays[0], self.rays[1], gt_c2w, True)
        # the batch's ONE D2H, behind an event instead of a device sync: [c2w (16) | select status | solve status | t err | ang err | mean kept weight | kept]
        full = torch.cat([sol["packed"], sol["status"].to(torch.float32)[:, None], sol["errors"].to(torch.float32),
                          (sol["w_final"].sum(dim=1) / sol["n_kept"].clamp(min=1))[:, None], sol["n_kept"].to(torch.float32)[:, None]], dim=1)
        host = torch.empty(full.shape, dtype=full.dtype, pin_memory=True)
        host.copy_(full, non_blocking=True)
        done = torch.cuda.Event()
        done.record(main)
        return {"sol": sol, "host": host, "done": done}

    @torch.no_grad()
    def collect(self, handle):
        """-> (c2w [B,4,4] on the host, sol).  Images the select path refused are re-done here by the two-pass scorer (rare)."""
        handle["done"].synchronize()
        return resolve_poses(self.idm, handle["sol"], handle["host"][:, :17]), handle["sol"]

    @torch.no_grad()
    def collect_eval(self, handle):
        """collect() plus what the evaluation loop reports per image, all from the batch's one D2H:
        -> (c2w [B,4,4], solve status [B] int, errors [B,2] (translation, angular), mean kept weight [B], kept rays [B] int) on the host."""
        c2w, sol = self.collect(handle)
        h = handle["host"]
        if min(h[:, 16].round().to(torch.int64).tolist() or [0]) < 0:        # an image was re-solved by the two-pass path: read its values again (rare)
            st, err = sol["status"].cpu(), sol["errors"].cpu()
            wm, nk = (sol["w_final"].sum(dim=1) / sol["n_kept"].clamp(min=1)).cpu(), sol["n_kept"].cpu()
            return c2w, st, err, wm, nk
        return c2w, h[:, 17].round().to(torch.int32), h[:, 18:20].clone(), h[:, 20].clone(), h[:, 21].round().to(torch.int64)


@torch.no_grad()
def estimate_poses_ray_sharded(id_module, images, rays_ori, rays_dirs, rays_rgb, ray_offset: int, r_total: int, gt_c2w=None, k: int = 100,
                               profile=None, tokens=None, up=None, image_graph: bool = True, group=None):
    """estimate_poses for a scene whose rays are split across the ranks (this rank: rays [ray_offset, ray_offset + len(rays_ori)) of
    r_total, key planes resident): EVERY rank passes the same images, runs the (cheap) image side itself, scores its ray slice
    (IdentificationModule.score_tokens_ray_sharded), and the 100 selected rays of each image are put together from their owners
    (distributed.gather_selected_rays) for the pose solve, which every rank then runs identically."""
    from . import distributed as dd
    if tokens is None:
        res = None
        if image_graph and not torch.cuda.is_current_stream_capturing():
            cache = id_module.__dict__.setdefault("_image_side_graph", _ImageSideGraph())
            res = cache.run(id_module, images)
        if res is not None and not isinstance(res[0], (list, tuple)):
            tokens, up = res
        else:
            tokens, fmaps = image_side_tokens(id_module, images)
            up = id_module.camera_up(fmaps)
    idx, weights = id_module.score_tokens_ray_sharded(tokens, rays_ori, rays_dirs, rays_rgb, ray_offset, r_total, k, group=group, profile=profile)
    sel_o, sel_d = dd.gather_selected_rays(idx, rays_ori, rays_dirs, ray_offset, group)
    b = idx.shape[0]
    compact = torch.arange(b * k, device=idx.device, dtype=torch.int64).view(b, k)
    compact = torch.where(idx >= 0, compact, idx)
    sol = ops.solve_pose(sel_o.reshape(-1, 3), sel_d.reshape(-1, 3), compact, weights, up, gt_c2w)
    sol.update(idx=idx, weights=weights, scores=None, tokens=tokens, up=up)
    return sol


def gt_pose_and_intrinsics(camera_info, device):
    """test.py:47-67 (on the host: 4x4 inverse of [R^T | T])."""
    w2c = torch.eye(4, dtype=torch.float32)
    w2c[:3, :3] = torch.transpose(torch.from_numpy(np.asarray(camera_info.R)), -1, -2).float()
    w2c[:3, -1] = torch.from_numpy(np.asarray(camera_info.T)).float()
    c2w = torch.inverse(w2c)
    fx, fy = fov2focal(camera_info.FovX, camera_info.width), fov2focal(camera_info.FovY, camera_info.height)
    K = torch.tensor([[fx, 0.0, camera_info.width / 2], [0.0, fy, camera_info.height / 2], [0.0, 0.0, 1.0]], dtype=torch.float32)
    return c2w, K


def test_pose_estimation(
    cameras_info,
    id_module,
    rays_ori,
    rays_dirs,
    rays_rgb,
    model_up,
    sequence_id="",
    category_id="",
    loss_fn=None,
    save=False,
    save_all=False,
    *,
    batch_size: int = 16,
    verbose: bool = True,
    token_override: Optional[List[torch.Tensor]] = None,
    up_override: Optional[torch.Tensor] = None,
    save_dir: Optional[str] = None,
):
    """See module docstring.  `token_override` / `up_override` inject the boundary's image-side inputs
    (tokens [T,398] per image, camera-up [B,3]) -- used by the parity fixtures, where DINOv2 is absent.

    `save` / `save_all` (test.py:94-106,137-140,164-166,202-214): the per-image dump of everything the estimate was made from
    (`sample_results_<img_idx>.th`, image 0 only unless `save_all`), same keys as the reference.  The reference writes into a
    developer's home directory; here the directory is `save_dir`, else $SIXDGS_SAVE_DIR, else ./sample_results."""
    if save:
        save_dir = save_dir or os.environ.get("SIXDGS_SAVE_DIR") or os.path.join(os.getcwd(), "sample_results")
        os.makedirs(save_dir, exist_ok=True)
    id_module.eval()
    dev = rays_ori.device
    if not rays_ori.is_cuda:
        raise RuntimeError("6dgs_amd.test_pose_estimation needs the rays on the GPU (no CPU fallback)")
    model_up = torch.divide(model_up, torch.linalg.norm(model_up, dim=-1, keepdim=True))  # test.py:40 (unused afterwards)
    n = len(cameras_info)
    translation_errors, angular_errors, recalls, avg_loss_scores, results = [], [], [], [], []
    k = 100
    start_time = time.time()
    # batches of EQUAL size (19 views at batch_size 16: 10 + 9, not 16 + 3): the reference scores one image at a time, so the cut is free, and a short last
    # batch runs the sweep at the efficiency of a near-empty launch (one or two tiles pull every key tile from HBM for themselves)
    if n > 0 and batch_size > 0:
        batch_size = -(-n // (-(-n // batch_size)))

    def record(b0, i, gts_b, c2w, st, t_err, a_err, w_mean_i, nk_i, avg_score_i, recall_i):
        if verbose:
            if st & 4:
                print("camera_optical_center is nan")
            if st & 1:
                print("extracted rotation matrix is singular")
            if st & 2:
                print("wrong c2w")
        translation_errors.append(float(t_err))
        angular_errors.append(float(a_err))
        avg_loss_scores.append(avg_score_i)
        recalls.append(recall_i)
        results.append({
            "sequence_id": sequence_id,
            "category_name": category_id,
            "frame_id": b0 + i,
            "loss": float(w_mean_i) if int(nk_i) > 0 else float("nan"),
            "scores_loss": avg_score_i,
            "recall": recall_i,
            "total_optimization_time_in_ms": 0.0,
            "pred_c2w": c2w[i].tolist(),
            "gt_c2w": gts_b[i].tolist(),
        })

    # The inference pass (no loss_fn, nothing saved: test.py:85-107 reads only the top-k) as a PIPELINE of batches (PoseStream): the host decodes and
    # uploads batch N + 1 and its image side runs while batch N is being scored; one D2H per batch behind an event.  Same results as the loop below.
    streamable = (loss_fn is None and not save and os.environ.get("SIXDGS_NO_PIPELINE") != "1" and
                  (token_override is not None or all(np.asarray(c.image).dtype == np.uint8 for c in cameras_info)))
    if streamable and n > 0:
        ps = PoseStream(id_module, rays_ori, rays_dirs, rays_rgb, k)
        pending = None
        for b0 in list(range(0, n, batch_size)) + [None]:
            cur = None
            if b0 is not None:
                cams = cameras_info[b0:b0 + batch_size]
                gts, _ = zip(*[gt_pose_and_intrinsics(c, dev) for c in cams])
                with torch.cuda.stream(ps.image_stream):            # uploads on the image stream: they do not wait for the batch being scored
                    gt = torch.stack(gts).to(dev, non_blocking=False)
                    if token_override is not None:
                        toks = [t.to(dev) for t in token_override[b0:b0 + len(cams)]]
                        up = up_override[b0:b0 + len(cams)].to(dev)
                        ev = torch.cuda.Event()
                        ev.record(ps.image_stream)
                    else:
                        imgs = [torch.from_numpy(np.ascontiguousarray(np.array(c.image))).to(dev) for c in cams]
                if token_override is not None:
                    torch.cuda.current_stream().wait_event(ev)
                    for t in toks + [up, gt]:
                        t.record_stream(torch.cuda.current_stream())
                    cur = (b0, gts, ps.submit(None, gt, tokens=toks, up=up))
                else:
                    gt.record_stream(torch.cuda.current_stream())
                    cur = (b0, gts, ps.submit(imgs, gt))
            if pending is not None:
                pb0, pgts, handle = pending
                c2w, status, err, w_mean, nk = ps.collect_eval(handle)
                for i in range(c2w.shape[0]):
                    record(pb0, i, pgts, c2w, int(status[i]), err[i, 0], err[i, 1], w_mean[i], nk[i], -1.0, -1.0)
            pending = cur
        n_done = n
    else:
        n_done = 0
    for b0 in range(n_done, n, batch_size):
        cams = cameras_info[b0:b0 + batch_size]
        nb = len(cams)
        gts, Ks = zip(*[gt_pose_and_intrinsics(c, dev) for c in cams])
        gt = torch.stack(gts).to(dev)
        if token_override is not None:
            toks = [t.to(dev) for t in token_override[b0:b0 + nb]]
            up = up_override[b0:b0 + nb].to(dev)
        else:
            arrs = [np.asarray(c.image) for c in cams]
            has_alpha = [a.shape[-1] == 4 for a in arrs]
            rgb_u8 = all(a.dtype == np.uint8 and a.ndim == 3 and a.shape == arrs[0].shape and a.shape[-1] == 3 for a in arrs)
            will_save = bool(save) and (b0 == 0 or bool(save_all))
            prepared = None if (rgb_u8 and not will_save) else [prepare_image(c.image, dev) for c in cams]
            if rgb_u8:      # the same image side as the pipelined pass (one kernel from uint8 to normalised planes where it applies): same results bit for bit
                toks, fmaps = image_side_tokens(id_module, [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrs])
            else:
                toks, fmaps = id_module.image_tokens([p[0] for p in prepared], [p[1] if a else None for p, a in zip(prepared, has_alpha)])
            up = id_module.camera_up(fmaps)
        saving = [bool(save) and (b0 + i == 0 or bool(save_all)) for i in range(nb)]
        idx, weights, pred_scores = id_module.score_tokens(toks, rays_ori, rays_dirs, rays_rgb, k, want_scores=loss_fn is not None or any(saving))
        dumps = [None] * nb
        for i in range(nb):
            if saving[i]:       # test.py:94-106
                dumps[i] = {
                    "gt_pose": gt[i].cpu(), "camera_intrinsic": Ks[i].cpu(),
                    "all_rays_ori": rays_ori.cpu(), "all_rays_dirs": rays_dirs.cpu(), "all_rays_rgb": rays_rgb.cpu(),
                    "obs_img": None if token_override is not None else prepared[i][0].cpu(),
                    "mask_img": None if token_override is not None else prepared[i][1].cpu(),
                    "topk_nonunique_ray_idx": idx[i].cpu(), "topk_nonunique_weights": weights[i].cpu(),
                    "all_predict_weights": pred_scores[i].cpu(),
                }
        avg_score = [-1.0] * nb
        recall = [-1.0] * nb
        if loss_fn is not None:  # test.py:108-142: evaluate the GROUND-TRUTH top-k instead of the prediction
            new_idx, new_w = [], []
            for i in range(nb):
                s_i, target_scores = loss_fn(pred_scores[i], gt[i], Ks[i].to(dev), rays_ori, rays_dirs, int(toks[i].shape[0]),
                                             id_module.backbone_wrapper.backbone_wh, model_up=up[i])
                avg_score[i] = s_i.item()
                target_idx, _ = ops.topk(weights[i], k)
                recall[i] = torch.count_nonzero(torch.isin(target_idx, idx[i])).item() / target_idx.shape[0]
                ti, tw = ops.topk(target_scores, k)
                new_idx.append(ti)
                new_w.append(tw)
                if saving[i]:   # test.py:137-140
                    dumps[i].update(all_target_weights=target_scores.cpu(), loss=avg_score[i], recall=recall[i])
            idx, weights = torch.stack(new_idx), torch.stack(new_w)
        sol = ops.solve_pose(rays_ori, rays_dirs, idx, weights, up, gt)
        c2w = sol["c2w"].cpu()
        status = sol["status"].cpu()
        err = sol["errors"].cpu()
        w_mean = (sol["w_final"].sum(dim=1) / sol["n_kept"].clamp(min=1)).cpu()  # weights.mean() over kept rays
        nk = sol["n_kept"].cpu()
        for i in range(nb):
            if saving[i]:       # test.py:157-166,202-214: the duplicate-origin filter restated on the 100 selected rays, for the dump only
                sel_o = rays_ori[idx[i]]
                uniq, counts = torch.unique(sel_o, return_counts=True, dim=0)
                keep = torch.isin(sel_o, uniq[counts == 1], assume_unique=True).any(dim=1)
                wf = sol["w_final"][i]
                watch = torch.multiply(rays_dirs[idx[i]], wf[:, None]).sum(dim=0)
                watch = torch.divide(watch, torch.linalg.norm(watch, dim=-1, keepdim=True))
                dumps[i].update(topk_unique_ray_idx=idx[i][keep].cpu(), topk_unique_weights=weights[i][keep].cpu(),
                                topk_unique_weights_after_exclusion=wf[keep].cpu(), pred_camera_optical_center=sol["centre"][i].cpu(),
                                pred_camera_watch_dir=(-watch).cpu(), pred_c2w_matrix=c2w[i].clone(), model_up=model_up.cpu())
                torch.save(dumps[i], os.path.join(save_dir, f"sample_results_{b0 + i}.th"))
                if verbose:
                    print("Sample result saved")
            record(b0, i, gts, c2w, int(status[i]), err[i, 0], err[i, 1], w_mean[i], nk[i], avg_score[i], recall[i])
    total_time = time.time() - start_time
    time_per_element = total_time / max(n, 1)
    avg_loss_score = mean(avg_loss_scores) if avg_loss_scores else float("nan")
    avg_recall = mean(recalls) if recalls else float("nan")
    avg_translation_error = mean(translation_errors) if translation_errors else float("nan")
    avg_angular_error = mean(angular_errors) if angular_errors else float("nan")
    if verbose:
        print("Average loss score: ", avg_loss_score)
        print("Average Recall: ", avg_recall)
        print("Time per element: ", time_per_element)
        print("Translation Error: ", avg_translation_error)
        print("Angular Error: ", avg_angular_error)
        if translation_errors:
            print("Smallest translation error: ", min(range(len(translation_errors)), key=translation_errors.__getitem__))
    return results, avg_translation_error, avg_angular_error, avg_loss_score, avg_recall


test_pose_estimation.__test__ = False  # not a pytest test
