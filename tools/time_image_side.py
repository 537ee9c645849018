"""GPU time of the image side of a batch as the product replays it (test._ImageSideGraph: uint8 -> planar fp32, resize, ViT-S/14, token assembly as one
hipGraph; the camera-up CNN as a second), nothing else on the device, for 1 / 4 / 8 / 16 images of 800 x 800.  SIXDGS_VIT_FUSED=0 gives PyTorch's kernels
for the ViT blocks.  python tools/time_image_side.py"""
import importlib
import os
import sys

os.environ.setdefault("SIXDGS_RANDOM_BACKBONE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

pkg = importlib.import_module("6dgs_amd")
syn = importlib.import_module("6dgs_amd.synthetic")
tp = importlib.import_module("6dgs_amd.test")


def main():
    dev = torch.device("cuda", 0)
    idm = pkg.IdentificationModule("dino")
    idm.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_scorer_state_dict(0, with_cnn=True).items()}, strict=False)
    idm = idm.to(dev).eval()
    print("| images | ViT graph (prep + resize + ViT + tokens) us | camera-up CNN graph us | image side us | per image us |")
    print("|---:|---:|---:|---:|---:|")
    for n in (1, 4, 8, 16):
        cams = syn.make_cameras(n, 100, width=800, height=800)
        images = [torch.from_numpy(c["image"]).to(dev) for c in cams]
        cache = tp._ImageSideGraph()
        with torch.no_grad():
            assert cache.run(idm, images) is not None
        torch.cuda.synchronize()
        ev = lambda: torch.cuda.Event(enable_timing=True)
        a, b, c = ev(), ev(), ev()
        reps = 50
        tv = tc = 0.0
        for _ in range(reps):
            a.record()
            cache.g_vit.replay()
            b.record()
            cache.g_cnn.replay()
            c.record()
            torch.cuda.synchronize()
            tv += a.elapsed_time(b)
            tc += b.elapsed_time(c)
        tv, tc = tv / reps * 1e3, tc / reps * 1e3
        print(f"| {n} | {tv:.0f} | {tc:.0f} | {tv + tc:.0f} | {(tv + tc) / n:.0f} |")


if __name__ == "__main__":
    main()
