"""Times the iso-cell emitter at the headline size (500 k Gaussians x 64 rays) with HIP events; prints ms and GB/s of the 36 B/ray written."""
import importlib, os, sys
os.environ.setdefault("SIXDGS_RANDOM_BACKBONE", "1")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("6dgs_amd"); syn = importlib.import_module("6dgs_amd.synthetic"); ops = importlib.import_module("6dgs_amd.ops")
n, k = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000, int(sys.argv[2]) if len(sys.argv) > 2 else 64
scene = pkg.GaussianScene.from_dict(syn.make_scene(n, 0), device="cuda")
normals = ops.normals_knn(scene._xyz, scene._xyz, 20)
dirs = ops.isocell_distribution(k, 1, device="cuda")
for want_rgb in (True, False):
    ts = []
    for it in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = ops.emit_isocell(scene._xyz, scene._scaling, scene._rotation, scene._features_dc, scene._features_rest, 3 if want_rgb else 0, None, normals, dirs, want_rgb=want_rgb, want_src=False)
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms = sorted(ts[1:])[len(ts[1:]) // 2]          # includes the output allocation (cached after the first call)
    byt = n * k * (36 if want_rgb else 24) + n * 236
    print(f"emit_isocell N={n} K={k} rgb={want_rgb}: {ms:.3f} ms, {byt / ms / 1e6:.0f} GB/s")
