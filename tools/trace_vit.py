"""Replay the ViT-S/14 forward of `images` 224 x 224 images as a hipGraph `reps` times (for a rocprofv3 kernel trace of the image side).
python tools/trace_vit.py <images> <reps>      (SIXDGS_VIT_FUSED selects this build's blocks or PyTorch's kernels)"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

images, reps = int(sys.argv[1]), int(sys.argv[2])
bb = importlib.import_module("6dgs_amd.backbone")
torch.manual_seed(0)
vit = bb.ViTS14().eval().cuda()
x = torch.randn(images, 3, 224, 224, device="cuda")
with torch.no_grad():
    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        for _ in range(3):
            vit.forward_features(x)
    torch.cuda.current_stream().wait_stream(s_)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = vit.forward_features(x)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    print(f"images {images} fused {os.environ.get('SIXDGS_VIT_FUSED', '1')}: {(time.perf_counter() - t0) / reps * 1e6:.0f} us per replay")
