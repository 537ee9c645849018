"""CameraDirectionPredictor -- parameter-compatible mirror of
pose_estimation/camera_direction_network.py:6-90 (3x Conv5 + Conv4 + MLP 384->256->3 on the 16x16
DINOv2 feature map).  It stays on PyTorch-ROCm / MIOpen behind the boundary (SURVEY.md §8 a22:
~1.65 GFLOP per pose, not the bottleneck); module names match the reference's state_dict keys.
"""
from __future__ import annotations

from math import prod

import torch


class CameraDirectionPredictor(torch.nn.Module):
    def __init__(self, image_feature_channel=256, image_size=(16, 16), pospe=8, featureC=256, fea_output=3):
        super().__init__()
        self.direction_input = 2 * pospe * 3 + 3
        self.dim_reducer1, size1 = self._reducer(image_size, image_feature_channel, kernel_size=5, num_conv2d=3)
        self.dim_reducer2, size2 = self._reducer(size1, image_feature_channel, kernel_size=4, num_conv2d=1)
        self.in_mlpC = prod(size2) * image_feature_channel
        self.mlp = torch.nn.Sequential(
            torch.nn.Linear(self.in_mlpC, featureC), torch.nn.ReLU(inplace=True), torch.nn.Linear(featureC, fea_output))
        self.pospe = pospe

    @staticmethod
    def _reducer(image_size, ch, kernel_size, num_conv2d):
        layers = []
        size = list(image_size)
        for _ in range(num_conv2d):
            size = [int(s - kernel_size + 1) for s in size]
            layers += [torch.nn.Conv2d(ch, ch, kernel_size=kernel_size), torch.nn.ReLU(inplace=True)]
        return torch.nn.Sequential(*layers), size

    def forward(self, image_features):
        """image_features [384,16,16] -> [3]   or batched [B,384,16,16] -> [B,3]"""
        single = image_features.dim() == 3
        x = image_features[None] if single else image_features
        if x.is_cuda and not torch.is_grad_enabled():
            y = self._forward_gemm(x)
        else:
            x = self.dim_reducer2(self.dim_reducer1(x))
            y = self.mlp(x.reshape(x.shape[0], -1))
        return y[0] if single else y

    def _taps_major_weight(self, conv):
        """conv.weight [out, c, kh, kw] with its columns in (kh, kw, c) order, as ops.im2col(taps_major=True) lays the patches out: built once per
        parameter version (a side table, so the module's state_dict keeps the reference's keys)."""
        w = conv.weight
        cache = self.__dict__.setdefault("_tm_cache", {})
        hit = cache.get(id(conv))
        if hit is not None and hit[0] == (w._version, w.data_ptr(), str(w.device)):
            return hit[1]
        wt = w.detach().permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()
        if wt.is_cuda and not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(wt.device).synchronize()      # kept for the life of the weight and possibly read next from another stream: finish it here, once
        cache[id(conv)] = ((w._version, w.data_ptr(), str(w.device)), wt)
        return wt

    def _forward_gemm(self, x):
        """Inference on the GPU: each valid convolution of the 16x16 map is im2col (ops.im2col: the whole batch in one launch, the previous layer's
        GEMM output read in place through a permuted view; patches laid out taps-major, so that with the channel-contiguous maps of this path every
        patch row is a run of plain copies, against a weight whose columns are permuted to match) + the library's MFMA `linear` with the bias/ReLU
        epilogue.  MIOpen has no tuned fp32 solver for 384-channel 5x5 on 16x16 and falls back to naive_conv (2.7 ms per image in the round-1 trace);
        F.unfold launches one im2col kernel per image and needed a transposing copy (rounds 2-5: 1.1 ms of a 16-image step)."""
        from . import ops

        b = x.shape[0]
        for seq in (self.dim_reducer1, self.dim_reducer2):
            for layer in seq:
                if not isinstance(layer, torch.nn.Conv2d):
                    continue
                k = layer.kernel_size[0]
                ho, wo = x.shape[2] - k + 1, x.shape[3] - k + 1
                y = ops.linear(ops.im2col(x, k, taps_major=True), self._taps_major_weight(layer), layer.bias, relu=True)      # [b * ho * wo, out]
                x = y.view(b, ho, wo, -1).permute(0, 3, 1, 2)                                                                # [b, out, ho, wo], no copy
        h = ops.linear(x.reshape(b, -1).contiguous(), self.mlp[0].weight, self.mlp[0].bias, relu=True)
        return ops.linear(h, self.mlp[2].weight, self.mlp[2].bias, relu=False)
