"""Round 6: why did cfg-2's key-plane buffer (29.5 GB) take 3.7 s to allocate when the headline's 49 GB took 0.0 s?
Replays bench.py --config cfg2's set-up stage by stage and times a fresh allocation of the plane buffer after stage N (one process per N:
an allocation changes the allocator's state for the next).  python tools/probe_alloc.py --stage N [--gb 29.5]"""
import argparse
import ctypes
import importlib
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SIXDGS_RANDOM_BACKBONE", "1")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", type=int, default=0)
    ap.add_argument("--gb", type=float, default=29.5)
    ap.add_argument("--gaussians", type=int, default=300000)
    ap.add_argument("--raw", action="store_true", help="hipMalloc through ctypes instead of torch.empty")
    a = ap.parse_args()
    import torch
    pkg = importlib.import_module("6dgs_amd")
    syn = importlib.import_module("6dgs_amd.synthetic")
    dev = torch.device("cuda", 0)
    torch.cuda.init()
    torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    names = ["fresh context", "scene uploaded", "PLY written and read back", "module on the device", "rays emitted"]
    scene = None
    if a.stage >= 1:
        scene = pkg.GaussianScene.from_dict(syn.make_scene(a.gaussians, 0), device=dev)
    if a.stage >= 2:
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "point_cloud", "iteration_30000", "point_cloud.ply")
            scene.save_ply(path)
            scene = pkg.GaussianScene.load_ply(path, sh_degree=3, device=dev)
    if a.stage >= 3:
        idm = pkg.IdentificationModule("dino")
        idm.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_scorer_state_dict(0, with_cnn=True).items()}, strict=False)
        idm = idm.to(dev).eval()
    if a.stage >= 4:
        ori, dr, rgb = pkg.generate_all_possible_rays(scene, max_ellipsoids=-1, emitter="isocell", rays_per_ellipsoid=64)
    torch.cuda.synchronize()
    n = int(a.gb * 1e9)
    free0 = torch.cuda.mem_get_info(dev)[0]
    t0 = time.perf_counter()
    if a.raw:
        hip = ctypes.CDLL("libamdhip64.so")
        p = ctypes.c_void_p()
        rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n))
        t1 = time.perf_counter()
        rc2 = hip.hipMemsetAsync(p, 0, ctypes.c_size_t(n), None)
        hip.hipDeviceSynchronize()
        t2 = time.perf_counter()
        print(f"stage {a.stage} ({names[a.stage]}): raw hipMalloc {a.gb} GB rc={rc} {t1 - t0:.3f} s; memset of all of it {t2 - t1:.3f} s (rc {rc2}); free before {free0 / 2**30:.1f} GiB", flush=True)
        return
    x = torch.empty(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    x[:: 1 << 21].zero_()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    x.zero_()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    del x
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    y = torch.empty(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    print(f"stage {a.stage} ({names[a.stage]}): torch.empty {a.gb} GB {t1 - t0:.3f} s; touch one byte per 2 MB {t2 - t1:.3f} s; zero all {t3 - t2:.3f} s; "
          f"empty_cache {t4 - t3:.3f} s; second torch.empty {t5 - t4:.3f} s; free before {free0 / 2**30:.1f} GiB", flush=True)


if __name__ == "__main__":
    main()
