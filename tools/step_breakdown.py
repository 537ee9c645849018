"""Where one bench step goes, phase by phase (HIP events around the phases of test.estimate_poses)."""
import importlib, os, sys, time
import os
os.environ.setdefault("SIXDGS_RANDOM_BACKBONE", "1")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("6dgs_amd"); syn = importlib.import_module("6dgs_amd.synthetic")
ops = importlib.import_module("6dgs_amd.ops"); tp = importlib.import_module("6dgs_amd.test")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
MODE = sys.argv[2] if len(sys.argv) > 2 else "full"
BATCH = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda", 0)
scene = pkg.GaussianScene.from_dict(syn.make_scene(N, 0), device=dev)
idm = pkg.IdentificationModule("dino")
idm.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_scorer_state_dict(0, with_cnn=True).items()}, strict=False)
idm = idm.to(dev).eval()
if MODE == "full":
    ori, dr, rgb = pkg.generate_all_possible_rays(scene, max_ellipsoids=-1, emitter="isocell", rays_per_ellipsoid=64)
    fin = torch.isfinite(dr).all(dim=1)
    ori, dr, rgb = ori[fin].contiguous(), dr[fin].contiguous(), rgb[fin].contiguous()
else:
    ori, dr, rgb = pkg.generate_all_possible_rays(scene)
idm._ensure_keys(ori, dr, rgb)
R = ori.shape[0]
ws = torch.empty(ops.score_topk_workspace_bytes(R, min(BATCH, 2 if MODE == 'full' else BATCH), 100), dtype=torch.uint8, device=dev)
cams = syn.make_cameras(BATCH, 100, width=800, height=800)
images = [torch.from_numpy(c["image"]).to(dev) for c in cams]
ev = lambda: torch.cuda.Event(enable_timing=True)
for it in range(3):
    e = [ev() for _ in range(7)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e[0].record()
    imgs_f, masks = tp.prepare_images_device(images); e[1].record()
    tokens, fmaps = idm.image_tokens(imgs_f, masks); e[2].record()
    up = idm.camera_up(fmaps); e[3].record()
    idx, weights, scores = idm.score_tokens(tokens, ori, dr, rgb, 100, workspace=ws); e[4].record()
    sol = ops.solve_pose(ori, dr, idx, weights, up, None); e[5].record()
    host = sol["c2w"].cpu(); e[6].record()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
    names = ["image prep", "backbone tokens", "camera-up CNN", "q_proj + scorer + top-k", "pose solve", "D2H"]
    print(f"iter {it}: wall {wall:.2f} ms | " + " | ".join(f"{n} {e[i].elapsed_time(e[i + 1]):.2f}" for i, n in enumerate(names)))
