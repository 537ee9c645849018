#!/usr/bin/env python3
"""bench.py -- poses/sec of the MI355X-native 6DGS pose path on BASELINE.json's workloads.

  python bench.py --gpus N --steps K --warmup W [--config headline|cfg1|cfg2|cfg3|cfg4]

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank
per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), or started plainly -- then bench.py
re-launches itself under torch.distributed.run on 127.0.0.1 with a free port (one process per GPU over RCCL).

Workload (config.workload).  `--config headline` (default) is the configuration BASELINE.json's metric is quoted on:
synthetic 500 k-Gaussian scene (SURVEY.md §8(d) generator, seed 0), rays emitted from EVERY valid Gaussian with the
iso-cell emitter at 64 rays per ellipsoid (R = 32.0 M rays), 800x800 uint8 query images, 8 images per GPU per step (rounds 1-5: 4;
BASELINE.md's batches of 64 / 128 views over 8 GPUs are 8 / 16 per GPU; the 4-image figure stays in the line as `headline_b4`).
The other presets are BASELINE.json's `configs` entries (sizes per SURVEY.md §8), each image-sharded over the ranks:
  cfg1  10 k Gaussians x 64 rays, one 400x400 query                     (configs[0]; the reference's CPU-runnable case)
  cfg2  300 k Gaussians read back from a 3DGS PLY, x 64 rays, one query  (configs[1])
  cfg3  1 M Gaussians x 64 = 64 M rays, 8 images per GPU                  (configs[2]: 64 views over 8 GPUs)
  cfg4  2 M Gaussians x 256 = 512 M rays, 16 images per GPU, key cache does not fit: streamed scorer   (configs[3])
  cfg5-standin  the reference's 12-scene evaluation sweep at ITS scale with synthetic stand-ins (configs[4]: 0.3-6.1 M Gaussians per scene, 16-48
        test views each, masked RGBA views for the Tanks&Temples scenes; per-scene poses/s, one view per scene oracle-checked): tools/cfg5_standin.py
One step = one pass of the hot path over one batch: image prep -> ViT-S/14 tokens + camera-up CNN (PyTorch-ROCm,
random init: DINOv2 weights are not downloadable) -> q_proj -> ray<->token scorer over the cached key planes -> top-100
-> pose solve -> c2w on the host.  Scene set-up (normals, emission, ray MLP + k_proj key cache) happens once per
scene, as in the reference (pretrain_eval_attention.py:89), and is reported separately.

Timing: W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize, max over ranks;
value = (N * batch * K) / time.  Every step ends with the poses on the host.  By default the steps are PIPELINED (round 5: the
evaluation is a stream of batches, test.py:46-302): batch N + 1 is submitted -- its image side on a second stream -- before
batch N's poses are collected (6dgs_amd.test.PoseStream; same poses bit for bit; `--no-pipeline` = one batch at a time with a
device sync per step, rounds 1-4); `median_step` is then the median interval between the completions of consecutive batches
(SURVEY §8(d)), `config.pipeline` says which form ran.  Images and scene arrays are resident in HBM when the clock starts.  Rank 0 prints ONE JSON line with `roofline` (the logits kernel: algorithmic FLOP /
HIP-event time), `fp32_logits_mode` (the same workload with the logits kept in fp32 between the passes instead of
24-bit fixed point: the cost of not narrowing) and, at N = 1, `cpu_baseline` (the CPU oracle on a bounded ray sample).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("SIXDGS_RANDOM_BACKBONE", "1")   # synthetic benchmark: random-init ViT-S/14 (no network for the DINOv2 weights)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver

PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X dense fp32 MFMA peak (MI355X_MICROARCH.md, chip-level parameters)
PEAK_16BIT_MFMA_TFLOPS = 2500.0  # MI355X dense bf16/fp16 MFMA peak (same table); a logits kernel that spends n 16-bit MFMA terms
                                 # per fp32 product has the roofline 2500 / n in ALGORITHMIC (fp32-equivalent) FLOP/s

PRESETS = {
    #            Gaussians  rays/ellipsoid  images/GPU/step  query size  scene source        scorer
    "headline": dict(gaussians=500_000, rays_per_ellipsoid=64, batch=8, image_size=800, scene="synthetic", scoring="resident"),
    "cfg1": dict(gaussians=10_000, rays_per_ellipsoid=64, batch=1, image_size=400, scene="synthetic", scoring="resident"),
    "cfg2": dict(gaussians=300_000, rays_per_ellipsoid=64, batch=1, image_size=800, scene="ply", scoring="resident"),
    "cfg3": dict(gaussians=1_000_000, rays_per_ellipsoid=64, batch=8, image_size=800, scene="synthetic", scoring="resident"),
    "cfg4": dict(gaussians=2_000_000, rays_per_ellipsoid=256, batch=16, image_size=800, scene="synthetic", scoring="streamed"),
    # configs[4] at its scale with synthetic stand-ins: 12 scenes at the Mip-NeRF360 / Tanks&Temples sizes, all test views (tools/cfg5_standin.py)
    "cfg5-standin": dict(gaussians=0, rays_per_ellipsoid=64, batch=16, image_size=800, scene="synthetic", scoring="resident"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 5; 1 for cfg4)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(PRESETS), default="headline", help="BASELINE.json workload preset (see the module docstring)")
    ap.add_argument("--batch", type=int, default=None, help="query images per GPU per step (overrides the preset)")
    ap.add_argument("--gaussians", type=int, default=None)
    ap.add_argument("--rays-per-ellipsoid", type=int, default=None)
    ap.add_argument("--image-size", type=int, default=None)
    ap.add_argument("--scene", choices=["synthetic", "ply"], default=None,
                    help="ply: the synthetic scene is written as a 3DGS point_cloud.ply and read back through GaussianScene.load_ply")
    ap.add_argument("--scoring", choices=["resident", "streamed"], default=None,
                    help="resident: key planes cached in HBM (1536 B/ray); streamed: ray chunks whose keys are computed, used and dropped")
    ap.add_argument("--chunk-rays", type=int, default=8_388_608, help="streamed scorer: rays per chunk")
    ap.add_argument("--parallelism", choices=["auto", "image", "ray"], default="auto",
                    help="image: query images shard over the ranks, every rank holds the whole scene (north_star; default).  ray: the RAYS shard "
                         "over the ranks, every rank keeps the key planes of its slice resident and all ranks score the same --batch images per "
                         "step -- for scenes whose key planes fit only across GPUs (cfg4 at N >= 4: 786 GB / N per GPU, no per-step ray MLP).  "
                         "auto: image, except for a streamed preset whose key planes fit the ranks' HBM together (then ray: measured 1.85x on one rank's share)")
    ap.add_argument("--mode", choices=["full", "reference"], default="full",
                    help="full: every Gaussian, iso-cell emitter (headline); reference: 1000-ellipsoid quadricell subsample")
    ap.add_argument("--mma", choices=["default", "bf16x6", "f16x3", "f16x3l32", "f32"], default="default",
                    help="matrix-core scheme of the logits kernel (default = the library's default mode)")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="images whose [256,R] logits are resident at once (0 = as many of the batch as fit in 60 %% of the free HBM)")
    ap.add_argument("--graph", action="store_true",
                    help="capture the whole per-batch path (image prep, ViT, CNN, scorer, pose solve) in one hipGraph and replay it "
                         "per step: for the launch-bound small-scene regime (--mode reference); kernel timing needs the eager path")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="one batch at a time, every step ending in a device sync (rounds 1-4).  Default: two batches in flight (6dgs_amd.test.PoseStream): "
                         "batch N + 1 is submitted -- its image side on a second stream -- before batch N's poses are collected; same poses bit for bit")
    ap.add_argument("--no-select", action="store_true",
                    help="score with the two-pass scorer (logits through HBM) instead of the select path (top-k without materialised logits)")
    ap.add_argument("--l32-steps", type=int, default=-1,
                    help="steps of the secondary fp32-logits measurement (-1 = min(steps, 3) when the main mode is f16x3; 0 = skip)")
    ap.add_argument("--cpu-sample-rays", type=int, default=1 << 22,
                    help="rays of the CPU-oracle sample (also the sample of parity_vs_oracle; the select path takes scenes from 2^20 rays): 2^22 = ~10 s of "
                         "oracle time on the 128-core host + as much for the PyTorch-CPU figure")
    ap.add_argument("--skip-reference-mode", action="store_true", help="skip the secondary reference-mode figure (1000-ellipsoid quadricell emission)")
    ap.add_argument("--skip-cpu-baseline", action="store_true")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic_latest.json"))
    ap.add_argument("--b8-steps", type=int, default=-1, help="headline: timed steps of the secondary figure at the other images-per-step setting (headline_b4; headline_b8 under --batch 4); -1 = as many poses as half of --steps, 0 = skip")
    ap.add_argument("--scenes", default="", help="cfg5-standin: comma-separated substrings of the scene names to run (default: all twelve)")
    ap.add_argument("--scale", type=float, default=1.0, help="cfg5-standin: multiply every scene's Gaussian count (tests run the sweep at 1/100)")
    ap.add_argument("--views-cap", type=int, default=0, help="cfg5-standin: at most this many test views per scene (0 = the reference's counts)")
    ap.add_argument("--masked-batch", type=int, default=32,
                    help="cfg5-standin: images per step of a MASKED resident scene (views keep 50-180 of 256 tokens: 2.3 quarters of 64 on average, so 32 views fill "
                         "19 tiles where 2 x 16 views fill 20; the reference scores one image at a time, the cut is free)")
    ap.add_argument("--streamed-batch", type=int, default=32, help="cfg5-standin: images per step of a streamed scene (one pass of the ray MLP serves them all)")
    ap.add_argument("--no-arena", action="store_true", help="cfg5-standin: big per-scene buffers from PyTorch's caching allocator with empty_cache() between scenes (round 4) instead of one arena")
    ap.add_argument("--stream-above-rays", type=int, default=0,
                    help="cfg5-standin: score scenes with more rays than this streamed even when their key planes would fit (0 = by HBM size only); lets a "
                         "reduced-scale run exercise the streamed class")
    ap.add_argument("--oracle-rays", type=int, default=1 << 20, help="cfg5-standin: rays of the per-scene oracle check (a prefix of the scene)")
    args = ap.parse_args()
    for k, v in PRESETS[args.config].items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    if args.steps is None:
        args.steps = 1 if args.scoring == "streamed" else 5
    return args


def self_launch(args) -> None:
    """`python bench.py --gpus N` with N > 1 and no rank environment: become the launcher -- one process per GPU under
    torch.distributed.run, rendezvous on 127.0.0.1 (the container hostname may not resolve) on a free port."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    if os.environ.get("SIXDGS_BENCH_FORCE_DEVICE") is not None and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # test hook: several ranks share ONE GPU.  Every process would open 4 hardware queues (+ SDMA): eight of them oversubscribe the device's queue slots
        # and the scheduler starts preempting queues -- under which the runtime aborted 1 run in 40 with HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION inside a
        # stock PyTorch copy kernel (profiles/r06_eight_ranks_sigabrt.md).  One hardware queue per process keeps the rehearsal inside what the device runs natively.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "1")
    import torch
    pkg = importlib.import_module("6dgs_amd")
    syn = importlib.import_module("6dgs_amd.synthetic")
    dd = importlib.import_module("6dgs_amd.distributed")
    ops = importlib.import_module("6dgs_amd.ops")
    tp = importlib.import_module("6dgs_amd.test")
    ops.set_mma_mode({"default": ops.MMA_DEFAULT, "bf16x6": ops.MMA_BF16X6, "f16x3": ops.MMA_F16X3, "f16x3l32": ops.MMA_F16X3_L32, "f32": ops.MMA_F32}[args.mma])
    ops.set_select_enabled(not args.no_select)

    # test hooks (tests/ and CI only): run the N > 1 code path with several ranks on one device over gloo
    forced_dev = os.environ.get("SIXDGS_BENCH_FORCE_DEVICE")
    rank, world, local = dd.init_from_env(os.environ.get("SIXDGS_BENCH_BACKEND"), set_device=forced_dev is None)
    if forced_dev is not None:
        local = int(forced_dev)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback on the product path)")
    if local >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dd.warm_long_wait_group(dev)              # (exists already unless init_from_env left it to us: nccl with a forced device)
    torch.manual_seed(0)
    if args.config == "cfg5-standin":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import cfg5_standin
        idm = pkg.IdentificationModule("dino")
        idm.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_scorer_state_dict(0, with_cnn=True).items()}, strict=False)
        idm = idm.to(dev).eval()
        dd.broadcast_module(idm, 0)
        out = cfg5_standin.run(args, torch, pkg, syn, dd, ops, tp, idm, dev, rank, world, PEAK_16BIT_MFMA_TFLOPS)
        if rank == 0:
            print(json.dumps(out), flush=True)
        dd.barrier()
        return
    ranks_seen = dd.ranks_seen(dev)          # an all-reduce of ones over the process group (RCCL when world > 1)

    # ---- scene: rank 0 owns it, RCCL broadcast of the Gaussian arrays, local re-emission ------------------
    # Set-up is itemised (VERDICT r4 #3a): every item ends in a device sync and the items add up to scene_setup_s.total.  "standin_*" items exist only
    # because the scene is synthesised here; the reference reads a PLY (pretrain_eval_attention.py:89).
    t_setup = time.time()
    setup_items, t_lap = {}, [time.time()]

    def lap(name):
        torch.cuda.synchronize()
        now = time.time()
        setup_items[name] = round(setup_items.get(name, 0.0) + now - t_lap[0], 3)
        t_lap[0] = now

    scene = None
    host_scene = syn.make_scene(args.gaussians, 0) if rank == 0 else None
    lap("standin_host_scene_generation")
    if rank == 0:
        scene = pkg.GaussianScene.from_dict(host_scene, device=dev)
        if args.scene == "ply":                # the on-disk format either side of the path: 3DGS point_cloud.ply (gaussian_model.py:284-420)
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, "point_cloud", "iteration_30000", "point_cloud.ply")
                scene.save_ply(path)
                lap("standin_ply_write")
                scene = pkg.GaussianScene.load_ply(path, sh_degree=3, device=dev)
    del host_scene
    scene = dd.broadcast_scene(scene, 0, device=dev)
    lap("scene_read_upload_and_broadcast")
    idm = pkg.IdentificationModule("dino")
    idm.load_state_dict({k: torch.from_numpy(v) for k, v in syn.make_scorer_state_dict(0, with_cnn=True).items()}, strict=False)
    idm = idm.to(dev).eval()
    dd.broadcast_module(idm, 0)
    lap("module_weights_init_upload_and_broadcast")
    t0 = time.time()
    if args.parallelism == "auto":
        planes_per_rank = args.gaussians * args.rays_per_ellipsoid * 1536 / max(world, 1)
        fits_across = world > 1 and planes_per_rank < 0.6 * torch.cuda.get_device_properties(dev).total_memory
        args.parallelism = "ray" if (args.scoring == "streamed" and args.mode == "full" and fits_across) else "image"
    ray_sharded = args.parallelism == "ray"
    # (--graph: the whole per-batch path -- image side, q_proj, select path with its status read deferred to the step's one D2H, pose
    # solve -- replayed as ONE hipGraph.  Measured in round 3 at 1 image per step: 26.4 ms against 13.9 ms eager with the image side alone
    # as a graph -- replaying the scorer's ~60 nodes costs more than launching them -- so it stays opt-in.)
    if ray_sharded and args.mode != "full":
        raise SystemExit("--parallelism ray needs --mode full")
    if ray_sharded:
        args.scoring = "resident"              # the point of ray sharding: every rank's slice of the key planes stays in HBM
    if args.mode == "full":
        ori, dr, rgb = pkg.generate_all_possible_rays(scene, max_ellipsoids=-1, emitter="isocell",
                                                      rays_per_ellipsoid=args.rays_per_ellipsoid, shard=(rank, world) if ray_sharded else None)
        finite = torch.isfinite(dr).all(dim=1)   # normals exactly (anti)parallel to z give NaN rays (isocell.py:208-212)
        if not bool(finite.all()):
            ori, dr, rgb = ori[finite].contiguous(), dr[finite].contiguous(), rgb[finite].contiguous()
        del finite
    else:
        ori, dr, rgb = pkg.generate_all_possible_rays(scene)
    torch.cuda.synchronize()
    t_emit = time.time() - t0
    lap("normals_knn_and_emission")
    R = int(ori.shape[0])
    ray_offset, R_total = 0, R
    if ray_sharded and dd.is_dist():            # this rank's rays are [ray_offset, ray_offset + R) of R_total
        counts = dd.all_counts(R, dev)
        ray_offset, R_total = sum(counts[:rank]), sum(counts)
    streamed = args.scoring == "streamed"
    # resident key planes (1536 B per ray) + the ray-MLP chain's transient workspace at its smallest chunk (it shrinks its chunk when memory is tight)
    if not streamed and R * 1536 + ops.ray_keys_workspace_bytes(R, ops.RAY_KEYS_CHUNK_MIN) > 0.9 * torch.cuda.mem_get_info(dev)[0]:
        raise SystemExit(f"rank {rank}: the key planes of {R} rays ({R * 1536 / 2**30:.0f} GiB) do not fit this GPU: use --scoring streamed, or --parallelism ray on more GPUs")
    kprof = ops.KernelProfile()
    t0 = time.time()
    ws, inflight, k_ms, k_fl = None, args.batch, 0.0, 0.0
    t_alloc = 0.0
    if not streamed:
        # The key planes are ONE buffer of 1536 B per ray (49 GB at 32 M rays, 98 GB at 64 M): its first allocation is a hipMalloc that
        # maps and clears the pages at ~50 GB/s -- 2 s of the 2.7 s that round 2 reported as "ray_mlp_keys" at 64 M rays were this, not
        # the ray MLP (tools/time_setup.py).  It is paid once per process: PyTorch's caching allocator hands the block to the next
        # scene's planes.  Timed on its own here, then released to the allocator, so that ray_mlp_keys below is the chain itself.
        torch.cuda.synchronize()
        ta = time.time()
        tmp = torch.empty(R, 1536, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        t_alloc = time.time() - ta
        del tmp
        lap("key_plane_buffer_alloc")
        t0 = time.time()
        idm._ensure_keys(ori, dr, rgb, profile=kprof, sample_min_rays=max(4096, ops.SELECT_MIN_RAYS // world) if ray_sharded else None)
        torch.cuda.synchronize()
        k_ms, k_fl, _, _ = kprof.collect()
        lap("ray_mlp_key_planes_and_sample")
    t_keys = time.time() - t0
    if not streamed:
        # images whose [256, R] logits are resident at once (they share every key tile through L2): --in-flight, or as many
        # of the batch as fit in 60 % of the free HBM (784 B per ray and image next to the 1536 B/ray key planes)
        if args.in_flight > 0:
            inflight = min(args.in_flight, args.batch)
        else:
            free_b = torch.cuda.mem_get_info(dev)[0]
            while inflight > 1 and ops.score_topk_workspace_bytes(R, inflight, 100) > 0.6 * free_b:
                inflight -= 1
    # the select path (top-k without materialised logits) serves the timed steps when the scene has a ray sample; the two-pass
    # workspace (784 B per ray and image) is then only needed for its fallback and for the secondary two-pass figures
    use_select = (not streamed and ops.select_enabled() and idm._key_cache is not None and idm._key_cache.get("sample") is not None)
    if ray_sharded:
        use_select = False                      # (the ray-sharded scorer decides select / two-pass itself, all ranks together)

    def two_pass_ws():
        nonlocal ws
        if ws is None:
            # sized for fp32 logits (1024 B per ray and image) so that the fp32-logits figure runs in the same buffer; the 24-bit
            # mode (784 B) then has room to spare
            ws = torch.empty(ops.score_topk_workspace_bytes(R, inflight, 100), dtype=torch.uint8, device=dev)
        return ws

    if not streamed and not use_select and not ray_sharded:
        two_pass_ws()
    lap("scorer_workspace")
    t_setup = time.time() - t_setup

    # ---- query images resident on the device ------------------------------------------------------------------
    # image-sharded: every rank its own images; ray-sharded: every rank the SAME images (each scores them against its ray slice)
    cams = syn.make_cameras(args.batch, 100 + (0 if ray_sharded else rank), width=args.image_size, height=args.image_size)
    images = [torch.from_numpy(c["image"]).to(dev) for c in cams]
    gts = torch.stack([tp.gt_pose_and_intrinsics(pkg.CameraInfo(**c), dev)[0] for c in cams]).to(dev)
    tp.prime_image_graph(idm, images)        # set-up, like the key cache: the hipGraph of the image side for this batch shape
    torch.cuda.synchronize()
    prof = ops.KernelProfile()

    graph, graph_sol = None, None

    def run_batch(p):
        if ray_sharded:
            return tp.estimate_poses_ray_sharded(idm, images, ori, dr, rgb, ray_offset, R_total, gt_c2w=gts, profile=p)
        return tp.estimate_poses(idm, images, ori, dr, rgb, gt_c2w=gts, workspace=ws, profile=p,
                                 streamed_chunk_rays=args.chunk_rays if streamed else None, defer_status=bool(args.graph) or use_select)

    def step(p):
        if graph is not None:
            graph.replay()
            sol = graph_sol
        else:
            sol = run_batch(p)
        if ray_sharded:                                          # every rank solved the same poses: nothing to gather
            return sol["c2w"].cpu(), sol
        if "packed" in sol:                                      # graph / deferred status: poses + select statuses in ONE D2H
            c2w_local = tp.resolve_poses(idm, sol, sol["packed"].cpu())
            if not dd.is_dist():
                return c2w_local, sol
            c2w, st = dd.gather_poses(c2w_local.to(dev), sol["status"], 0, counts=[args.batch] * world)
            return (c2w if c2w is not None else c2w_local).cpu(), sol
        c2w, st = dd.gather_poses(sol["c2w"], sol["status"], 0, counts=[args.batch] * world)      # one fixed-size gather to rank 0
        host = (c2w if c2w is not None else sol["c2w"]).cpu()   # all poses on the host = end of the step
        return host, sol

    if args.graph:
        # The path is sync-free and works on caller-provided buffers, so the whole batch is one capturable stream of
        # launches: ~300 kernel launches (ViT blocks, im2col GEMMs, scorer, top-k, solve) become one hipGraphLaunch.
        step(None)                                    # eager once: lazy initialisation (weights packing, key cache, workspaces)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run_batch(None)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            graph_sol = run_batch(None)
        graph = g

    # Pipelined steps (default where the path allows it: select scorer on resident planes, image-sharded, no whole-step graph): the evaluation is a
    # stream of batches (test.py:46-302), so batch N + 1 is SUBMITTED before batch N's poses are collected -- the host never sits between two batches
    # and the image side of N + 1 (own stream) overlaps the small serial kernels behind sweep N.  Every batch still ends with its poses on the host.
    # (--mode reference: the small-scene regime scores through the two-pass kernels -- no ray sample -- and pipelines the same way: its scorer is in
    # stream order on one workspace, and what the pipeline hides there is the host's sync and the D2H between two launch-bound image sides)
    pipelined = bool((use_select or args.mode == "reference") and not args.graph and not args.no_pipeline and not ray_sharded and not streamed)
    ps_main = tp.PoseStream(idm, ori, dr, rgb, workspace=ws) if pipelined else None
    last_host, rank_s = [None], [0.0]

    def timed(n_steps, p, ps=None, images=images, gts=gts, batch=args.batch):
        """exactly n_steps steps bracketed by barrier + synchronize; (max-over-ranks seconds, per-step seconds, last results).  Pipelined: per-step
        seconds are the intervals between the COMPLETIONS of consecutive batches (poses on the host); with several ranks the poses of all steps go
        to rank 0 in ONE fixed-size gather at the end (north_star: "a final gather") instead of one per step behind the next batch's sweep."""
        ps = ps if ps is not None else ps_main
        torch.cuda.synchronize()
        dd.barrier()
        per, t_begin = [], time.perf_counter()
        if ps is None or (args.mode != "reference" and (not use_select or not ops.select_enabled())):
            for _ in range(n_steps):
                t1 = time.perf_counter()
                host, s = step(p)
                per.append(time.perf_counter() - t1)
            last_host[0] = host
        else:
            prev, t1, local = None, t_begin, []
            for i in range(n_steps + 1):
                cur = ps.submit(images, gts, profile=p) if i < n_steps else None
                if prev is not None:
                    c2w_local, s = ps.collect(prev)
                    local.append(c2w_local)
                    t2 = time.perf_counter()
                    per.append(t2 - t1)
                    t1 = t2
                prev = cur
            host = local[-1]
            if dd.is_dist():
                allp, _ = dd.gather_poses(torch.cat(local).to(dev), None, 0, counts=[batch * n_steps] * world)
                if allp is not None:
                    host = allp.cpu().view(world, n_steps, batch, 4, 4)[:, -1].reshape(-1, 4, 4)
            last_host[0] = host
        torch.cuda.synchronize()
        rank_s[0] = time.perf_counter() - t_begin          # this rank's own time for its n_steps steps (before the closing barrier)
        dd.barrier()
        return dd.max_over_ranks(time.perf_counter() - t_begin, dev), per, s

    for _ in range(args.warmup):
        step(None)
        if ps_main is not None and (args.mode == "reference" or (use_select and ops.select_enabled())):      # ... and through the pipelined path itself (its stream, its first graph replay there, its pinned buffers)
            ps_main.collect(ps_main.submit(images, gts))
    elapsed, per_step, sol = timed(args.steps, None if args.graph else prof)
    per_rank_s = dd.all_floats(rank_s[0], dev)          # every rank's own time for the K steps: the line names the slowest rank
    if args.graph:        # HIP events cannot be read out of a captured graph: the kernel's duration comes from a few EAGER steps of the same batch
        g_keep, graph = graph, None
        for _ in range(min(args.steps, 5)):
            step(prof)
        graph = g_keep
    l_ms, l_fl, l_by, l_n = prof.collect()
    path = getattr(idm, "last_scoring_path", "two-pass")
    cand = list(getattr(idm, "last_select_candidates", [])) if use_select else None

    mode = ops.effective_mma_mode()
    mma_name = {ops.MMA_F32: "f32", ops.MMA_BF16X6: "bf16x6", ops.MMA_F16X3: "f16x3", ops.MMA_F16X3_L32: "f16x3l32"}
    # ---- the same workload through the two-pass scorer: 24-bit fixed-point logits between the passes, and fp32 logits (VERDICT r1:
    # keep the cost of not narrowing visible).  The select path above stores no logits at all and re-scores its candidates in fp32.
    l32, l24 = None, None
    n32 = (min(args.steps, 3) if args.l32_steps < 0 else args.l32_steps) if (mode == ops.MMA_F16X3 and not args.graph and not streamed and not ray_sharded) else 0
    if n32 > 0:
        two_pass_ws()
        ops.set_select_enabled(False)
        if use_select:
            step(None)
            e24, _, _ = timed(n32, None)
            l24 = {"mma": "f16x3", "value": round(world * args.batch * n32 / e24, 4), "unit": "poses/s", "steps": n32,
                   "ms_per_step": round(1e3 * e24 / n32, 3),
                   "note": "two-pass scorer, logits through HBM as 24-bit fixed point (784 B/ray/image): the round-1 headline path"}
        ops.set_mma_mode(ops.MMA_F16X3_L32)          # same key planes (the cache is keyed on the plane format), 1024 instead of
        step(None)                                   # 784 B of logits per ray and image: the library regroups the images in `ws`
        e32, per32, _ = timed(n32, None)
        ops.set_mma_mode(ops.MMA_DEFAULT if args.mma == "default" else ops.MMA_F16X3)
        ops.set_select_enabled(not args.no_select)
        l32 = {"mma": "f16x3l32", "value": round(world * args.batch * n32 / e32, 4), "unit": "poses/s", "steps": n32,
               "ms_per_step": round(1e3 * e32 / n32, 3),
               "note": "two-pass scorer, logits between the passes as fp32 (1024 B/ray/image): no intermediate below fp32"}

    img_ranks = 1 if ray_sharded else world          # ray-sharded: all ranks work on the same args.batch images of a step
    poses = img_ranks * args.batch * args.steps
    value = poses / elapsed
    med = statistics.median(per_step)
    out = {
        "metric": "poses/sec", "value": round(value, 4), "unit": "poses/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
        "scaling": "strong" if ray_sharded else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "median_step": {"ms": round(1e3 * med, 3), "poses_per_s": round(img_ranks * args.batch / med, 4), "n": len(per_step),
                        "min_ms": round(1e3 * min(per_step), 3), "max_ms": round(1e3 * max(per_step), 3),
                        "note": ("rank-0 interval between the completions of consecutive batches (poses of the batch on the host); two batches in flight"
                                 if pipelined else "rank-0 wall time of each timed step (every step ends with the poses on the host)")},
        "arithmetic": ("fp32 results; q.K^T and the ray MLP / k_proj as 2 power-of-two-scaled fp16 planes x 3 MFMA terms (q_proj, CNN: 3 bf16 planes x 6 "
                       "terms), fp32 accumulation (measured error <= that of the fp32 MFMA chain)"
                       + ("; select path: no logits stored, candidates re-scored in fp32 (nothing below fp32 on the path)" if ("select" in path and not path.startswith("streamed")) else
                          ("; logits travel between the two scorer passes as 24-bit fixed point, absolute error <= 2^-20 per logit "
                           "(fp32_logits_mode = the same run without that narrowing)" if mode == ops.MMA_F16X3 else ""))),
        "config": {
            "workload": (f"{args.config}: synthetic {args.gaussians}-Gaussian scene" + (" (through a 3DGS PLY file)" if args.scene == "ply" else "") + ", "
                         + (f"iso-cell emission from every valid Gaussian x {args.rays_per_ellipsoid} rays" if args.mode == "full"
                            else "reference-mode quadricell emission from 1000 sampled ellipsoids")
                         + f" (R={R_total} rays), {args.image_size}x{args.image_size} uint8 queries, 256 tokens x 384, top-100, "
                         + (f"{args.batch} images/step scored by all {world} ranks (each on its ray slice)" if ray_sharded else f"{args.batch} images/GPU/step")
                         + "; backbone ViT-S/14 + camera-up CNN random-init on PyTorch-ROCm"),
            "preset": args.config, "mode": args.mode, "gaussians": args.gaussians, "rays": R_total, "images_per_gpu_per_step": args.batch,
            "scoring": args.scoring, "images_per_logits_launch": inflight, "hip_graph": bool(args.graph),
            "pipeline": ("2 batches in flight (6dgs_amd.test.PoseStream): batch N + 1 submitted before batch N's poses are collected, its image side on a second "
                         "stream; scorer in order on one stream; one D2H per batch behind an event; --no-pipeline = one batch at a time" if pipelined
                         else "none: one batch at a time, a device sync per step"),
            # select path: how the library cut this batch into sweep launches -- (256-token tiles, images) per launch, from the library's own planner with
            # the token counts of the last batch (csrc/sweep_plan.h: launches of 8 tiles, images packed into tiles by token count; SIXDGS_SWEEP_MAX_IMAGES)
            "select_sweep_launches": getattr(idm, "last_select_launches", None),
            "parallelism": (f"ray-sharded x{world} (scene broadcast over RCCL, every rank emits and keeps the key planes of its block of ellipsoids: "
                            f"{R} of {R_total} rays on rank 0; per batch a few KB of all-reduce / all-gather: sample statistics, g_t, U_(k), candidates)"
                            if ray_sharded else f"image-sharded x{world} (scene broadcast over RCCL, local re-emission, pose gather)"),
            "mma": mma_name[mode],
        },
        "ranks_seen": ranks_seen, "backend": dd.backend_name(),
        "per_rank_ms_per_step": [round(1e3 * t / args.steps, 3) for t in per_rank_s], "slowest_rank": int(max(range(len(per_rank_s)), key=per_rank_s.__getitem__)),
        "scene_setup_s": {"total": round(t_setup, 3), "breakdown": setup_items,
                          "standin": round(sum(v for k_, v in setup_items.items() if k_.startswith("standin_")), 3), "normals+emission": round(t_emit, 3), "key_plane_buffer_alloc": round(t_alloc, 3),
                          "ray_mlp_keys": round(t_keys, 3), "ray_mlp_keys_kernels": round(k_ms * 1e-3, 3),
                          "ray_mlp_keys_tflops": round(k_fl / (k_ms * 1e-3) / 1e12, 2) if k_ms > 0 else None,
                          "note": "key_plane_buffer_alloc = first hipMalloc of the 1536 B/ray plane buffer (once per process, reused across scenes); "
                                  "ray_mlp_keys = wall time of the ray MLP + k_proj chain incl. the select path's ray sample; _kernels = its HIP-event time"},
    }
    if os.environ.get("SIXDGS_BENCH_DUMP_POSES"):        # test hook: the poses of the last timed step in the line
        out["poses_last_step"] = sol["c2w"].cpu().tolist()          # (the device tensor of the last batch: identical with and without the pipeline)
    if streamed:
        out["config"]["chunk_rays"] = args.chunk_rays
        out["config"]["scoring_note"] = ("key planes of the whole scene (1536 B/ray) exceed one GPU: ray chunks go through the ray MLP + k_proj and the scorer "
                                         + ("ONCE (select path: U of every ray kept, 4 B per ray and image; the candidates' keys are recomputed and re-scored exactly)"
                                            if path.startswith("streamed select") else "twice (row statistics, then scores + top-k merge)")
                                         + "; nothing of size R x 384 is resident")
    out["config"]["scoring_path"] = path
    tk = sol.get("tokens") if isinstance(sol, dict) else None
    if tk is not None:      # tokens that survive the wrapper's mask per image of the last step (256 for RGB queries): the roofline credits 2*T*384 FLOP per ray with THESE counts
        out["config"]["tokens_per_image"] = [int(tk[i].shape[0]) for i in range(len(tk))]
    if cand is not None:
        out["config"]["select_candidates_last_batch"] = cand
    # ---- secondary figure: the same scene at the OTHER images-per-step setting.  BASELINE.md's batches are 64 / 128 views over 8 GPUs = 8 / 16 per
    # GPU: since round 6 the headline runs 8 images per GPU and step (a sweep launch of 8 tiles shares every key tile 8 ways instead of 4 and the per-batch
    # tail is paid once per 8 poses; 16 = two such launches, measured equal: profiles/r06_images_per_step.md) and `headline_b4` is the 4-image figure
    # of rounds 1-5 for continuity (with --batch 4 the roles swap: `headline_b8`).  Same path, same pipeline.
    alt = 4 if args.batch == 8 else 8
    if args.config == "headline" and pipelined and args.b8_steps != 0 and args.mode == "full":
        camsa = syn.make_cameras(alt, 500 + rank, width=args.image_size, height=args.image_size)
        imagesa = [torch.from_numpy(c["image"]).to(dev) for c in camsa]
        gtsa = torch.stack([tp.gt_pose_and_intrinsics(pkg.CameraInfo(**c), dev)[0] for c in camsa]).to(dev)
        tp.prime_image_graph(idm, imagesa)
        psa = tp.PoseStream(idm, ori, dr, rgb)
        for _ in range(2):
            psa.collect(psa.submit(imagesa, gtsa))
        na = max(2, (args.steps * args.batch) // (2 * alt)) if args.b8_steps < 0 else args.b8_steps
        pa = ops.KernelProfile()
        ea, pera, _ = timed(na, pa, ps=psa, images=imagesa, gts=gtsa, batch=alt)
        ma, fa, _, ca = pa.collect()
        out["headline_b%d" % alt] = {"value": round(world * alt * na / ea, 4), "unit": "poses/s", "images_per_gpu_per_step": alt, "steps": na, "ms_per_step": round(1e3 * ea / na, 3),
                                     "median_step_ms": round(1e3 * statistics.median(pera), 3), "sweep_avg_launch_ms": round(ma / max(ca, 1), 4),
                                     "sweep_tflops": round(fa / (ma * 1e-3) / 1e12, 2) if ma > 0 else None,
                                     "note": ("the headline workload at %d images per GPU and step (one sweep launch of %d tiles), same pipeline" % (alt, alt))
                                             + ("; this is what `value` was in rounds 1-5" if alt == 4 else "")}
        del psa, imagesa, gtsa
        idm._select_ws = None
    if l24 is not None:
        out["two_pass_mode"] = l24
    if l32 is not None:
        out["fp32_logits_mode"] = l32
    if rank == 0:
        traffic, traffic_source = None, None
        if os.path.exists(args.traffic_json):
            try:
                tj = json.load(open(args.traffic_json))
                is_sel = "select" in path and not path.startswith("streamed")
                sl = out["config"].get("select_sweep_launches")      # (as recorded behind the timed steps: the secondary figure below it has run other batches since)
                per_launch = (sl[0][1] if (is_sel and sl) else inflight)      # images the dominant kernel's launch scores: the select sweep's plan, or the two-pass workspace's share
                if tj.get("rays") == R and tj.get("mode") == args.mode and tj.get("mma") == out["config"]["mma"] and tj.get("images_per_launch") == per_launch and tj.get("path", "two-pass") == ("select" if is_sel else "two-pass"):
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = ("NOT measured in this run: copied from the builder's separate rocprofv3 --pmc pass over this same command (round %s): %s"
                                      % (tj.get("round"), tj.get("source", os.path.relpath(args.traffic_json, ROOT))))
            except Exception:
                traffic = None
        ach = l_fl / (l_ms * 1e-3) / 1e12 if l_ms > 0 else 0.0
        terms = {ops.MMA_F32: 1, ops.MMA_BF16X6: 6, ops.MMA_F16X3: 3, ops.MMA_F16X3_L32: 3}[mode]
        peak = PEAK_F32_MFMA_TFLOPS if mode == ops.MMA_F32 else PEAK_16BIT_MFMA_TFLOPS / terms
        out["roofline"] = {
            "kernel": {ops.MMA_BF16X6: "k_logits<bf16x6>: q.K^T on fp32 keys, the operands split into 3 bf16 planes on the fly, 6 cross terms on v_mfma_f32_32x32x16_bf16 (fp32-equivalent result), online row stats, logits stored once",
                       ops.MMA_F16X3: ("k_logits_f16x<UB> (select path): K.Q^T (256 rays x 256 tokens per tile) with fp32 operands scaled by a power of two "
                                       "and split into 2 fp16 planes, 3 cross terms on v_mfma_f32_32x32x16_f16 (fp32-equivalent result), LDS-DMA "
                                       "rings; epilogue exp2 + token-sum butterfly, 16 B per ray and image leave the chip (no logits)"
                                       if ("select" in path and not path.startswith("streamed")) else
                                       "k_logits_f16x: q.K^T (256 tokens x 256 rays per tile) with fp32 operands scaled by a power of two and split "
                                       "into 2 fp16 planes, 3 cross terms on v_mfma_f32_32x32x16_f16 (fp32-equivalent result), LDS-DMA rings, "
                                       "online row stats, logits stored once as 24-bit fixed point"),
                       ops.MMA_F16X3_L32: "k_logits_f16x: as f16x3 with the logits stored as fp32",
                       ops.MMA_F32: "k_logits<f32>: q.K^T on v_mfma_f32_32x32x2_f32, online row stats, logits stored once"}[mode],
            "bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(ach / peak, 4),
            "peak_basis": "157.3 TFLOP/s dense fp32 MFMA" if mode == ops.MMA_F32 else
                          ("2500 TFLOP/s dense 16-bit MFMA / %d MFMA terms per fp32 product; achieved = algorithmic 2*T*384 FLOP per "
                           "ray and image / HIP-event time (executed 16-bit MFMA rate = %dx achieved)" % (terms, terms)),
            "achieved_vs_fp32_mfma_peak": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
            "launches": l_n, "avg_launch_ms": round(l_ms / max(l_n, 1), 4),
            "algorithmic_flop_per_launch": l_fl / max(l_n, 1), "algorithmic_bytes_per_launch": l_by / max(l_n, 1),
            "share_of_step_time": round((l_ms * 1e-3 / max(min(args.steps, 5) if args.graph else args.steps, 1)) / (elapsed / args.steps), 4),
        }
        if args.graph:
            out["roofline"]["timing_note"] = ("the timed steps replay ONE hipGraph per batch; the kernel's HIP-event duration was measured in %d eager steps of "
                                              "the same batch right after them" % min(args.steps, 5))
    # ---- secondary figure: the reference's own emission (1000 sampled ellipsoids, quadricell: R ~ 28.7 k), where the per-pose cost is
    # the image side (ViT + CNN) and launch overhead, not the scorer.  Every rank runs it (image-sharded like the headline).
    if args.mode == "full" and not args.skip_reference_mode and not streamed and not args.graph and not ray_sharded:
        out["reference_mode"] = reference_mode_figure(args, pkg, syn, tp, dd, idm, scene, dev, rank, world)
    if rank == 0:
        if world == 1 and not args.skip_cpu_baseline and not ray_sharded:
            out["cpu_baseline"], out["parity_vs_oracle"] = cpu_baseline(args, idm, ori, dr, rgb, R, sol, gts)
            # BASELINE.json's metric is "poses/sec ...; mean rot/trans err": the second half, answered from this line alone
            pv = out["parity_vs_oracle"]
            out["metric_errors"] = {"mean_rot_err_deg_vs_reference_outputs": pv["pose"]["rot_err_deg"],
                                    "mean_trans_err_vs_reference_outputs": pv["pose"]["trans_err"],
                                    "vs_synthetic_gt": pv["pose"]["vs_synthetic_gt"],
                                    "note": "HIP path vs the CPU oracle (= the reference's algorithm, pinned by tests/golden) on the same sample of the "
                                            "scene's rays, same image; the scorer weights are random-init, so the error against the synthetic ground "
                                            "truth says nothing about accuracy -- it is the same number on both paths, which is the point"}
        print(json.dumps(out), flush=True)
    dd.barrier()


def reference_mode_figure(args, pkg, syn, tp, dd, idm, scene, dev, rank, world, batch: int = 16, steps: int = 10):
    import torch
    torch.manual_seed(1)
    ori, dr, rgb = pkg.generate_all_possible_rays(scene)                       # sampling.py:127-267 defaults: 1000 ellipsoids, 50 cells
    cams = syn.make_cameras(batch, 300 + rank, width=args.image_size, height=args.image_size)
    images = [torch.from_numpy(c["image"]).to(dev) for c in cams]
    tp.prime_image_graph(idm, images)

    # pipelined like the headline loop (PoseStream: two batches in flight, one D2H per batch behind an event, one pose gather at the end)
    ps = tp.PoseStream(idm, ori, dr, rgb)
    ps.collect(ps.submit(images))
    torch.cuda.synchronize()
    dd.barrier()
    t0 = time.perf_counter()
    prev, local = None, []
    for i in range(steps + 1):
        cur = ps.submit(images) if i < steps else None
        if prev is not None:
            local.append(ps.collect(prev)[0])
        prev = cur
    if dd.is_dist():
        dd.gather_poses(torch.cat(local).to(dev), None, 0, counts=[batch * steps] * world)
    torch.cuda.synchronize()
    dd.barrier()
    el = dd.max_over_ranks(time.perf_counter() - t0, dev)
    return {"value": round(world * batch * steps / el, 2), "unit": "poses/s", "rays": int(ori.shape[0]), "images_per_gpu_per_step": batch, "steps": steps,
            "ms_per_step": round(1e3 * el / steps, 3),
            "note": "same scene, the reference's own emission (1000 randomly sampled ellipsoids, quadricell, 50 target cells): the workload the "
                    "reference's CPU path runs at ~3.7 poses/s; here the step is the image side (ViT-S/14 blocks through sixdgs_tok_linear / "
                    "sixdgs_tok_attention, camera-up CNN as batched im2col + GEMM), the scorer is ~0.5 ms of it; two batches in flight"}


def ops_mod():
    return importlib.import_module("6dgs_amd.ops")


def _pose_delta(a, b):
    """(rotation angle in degrees, translation distance) between two c2w matrices (the quantities of error_computation.py:3-8 for a pair)."""
    import numpy as np
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    # the angle of R_a R_b^T from the chord |R_a - R_b|_F = 2 sqrt(2) sin(angle / 2): well conditioned at small angles, where
    # acos((trace - 1) / 2) turns fp32 rounding of the matrices (1e-7) into 1e-2 degrees
    chord = np.linalg.norm(a[:3, :3] - b[:3, :3]) / (2.0 * np.sqrt(2.0))
    return float(np.degrees(2.0 * np.arcsin(min(1.0, chord)))), float(np.linalg.norm(a[:3, 3] - b[:3, 3]))


def cpu_baseline(args, idm, ori, dr, rgb, R, sol, gts):
    """The CPU oracle (oracle/sixdgs_oracle.c, OpenMP, all host cores) on a bounded sample of the same workload:
    the per-pose path (q_proj, 3-pass softmax scorer, top-100, pose tail) over the first `cpu_sample_rays` rays
    of the scene with the SAME keys, extrapolated linearly in R (the scorer is linear in R).

    Returns (cpu_baseline, parity_vs_oracle).  parity_vs_oracle: the HIP path (both scorers: the select path the timed steps took
    and the two-pass scorer) on THAT sample and image against what the oracle just computed -- the oracle is only the checker here,
    the HIP numbers come from the product path (IdentificationModule.score_tokens + sixdgs_solve_pose)."""
    import numpy as np
    import torch
    from oracle import oracle as O
    O.build()
    ops = ops_mod()
    rs = int(min(args.cpu_sample_rays, R))
    # the sample: every (R // rs)-th ray of the scene (round 4; rounds 2-3 took the first rs rays = the first eighth of the ellipsoids): rays are scored
    # independently of each other, so any subset is a scene of its own for both sides
    step_r = max(1, R // rs)
    sel_r = torch.arange(0, R, step_r, device=ori.device)[:rs]
    rs = int(sel_r.shape[0])
    o_s, d_s, c_s = ori.index_select(0, sel_r), dr.index_select(0, sel_r), rgb.index_select(0, sel_r)
    _, key = ops.ray_keys(o_s, d_s, c_s, idm.packed_weights(ori.device))
    key = key.cpu().numpy()
    o_np, d_np = o_s.cpu().numpy(), d_s.cpu().numpy()
    tok_dev = sol["tokens"][0].contiguous()
    tok = tok_dev.cpu().numpy()
    up = sol["up"][0].cpu().numpy()
    sd = {k: v.detach().cpu().numpy() for k, v in idm._scorer_params().items()}
    cores = O.num_threads()
    t0 = time.perf_counter()
    q = O.q_proj(tok, sd)
    s = O.attention_scores(q, key)
    idx, val = O.topk(s, 100)
    p_ref = O.pose_from_topk(o_np, d_np, idx, val, up)
    t = time.perf_counter() - t0
    per_pose = t * (R / rs)
    out = {"value": round(1.0 / per_pose, 6), "unit": "poses/s", "cores": cores, "kind": "port",
           "sample": f"per-pose path (q_proj + softmax scorer + top-100 + pose solve) on {rs} of {R} rays (every {step_r}-th ray of the scene), "
                     f"{t:.2f} s measured, scaled by R/sample; backbone/CNN excluded; scene set-up excluded",
           "sample_seconds": round(t, 3)}
    # The reference recomputes the ray MLP + k_proj for EVERY image (identification_module.py:79; this build caches the keys per scene):
    # the oracle's ray MLP on a slice of the sample, scaled to R, added to the per-pose figure above.
    try:
        rm = int(min(rs, 131072))
        t0 = time.perf_counter()
        _, okey = O.ray_features(o_np[:rm], d_np[:rm], c_s[:rm].cpu().numpy(), sd, want_feat=False)
        t_mlp = (time.perf_counter() - t0) * (R / rm)
        # ... and since the oracle's own keys of those rays now exist: the HIP ray MLP + k_proj against them (the scorer check below feeds the oracle the HIP keys)
        key_err = float((np.abs(key[:rm] - okey).max(axis=1) / np.abs(okey).max(axis=1)).max())
        out["reference_cost_per_pose"] = {"value": round(1.0 / (per_pose + t_mlp), 6), "unit": "poses/s",
                                          "note": f"as above PLUS the ray MLP + k_proj over all R rays per image, as the reference runs it "
                                                  f"(oracle on {rm} rays, scaled by R/{rm}: {t_mlp:.1f} s per pose)"}
    except Exception as e:
        out["reference_cost_per_pose"] = {"error": e.__class__.__name__}
    # ---- the HIP path on the same sample (its own key planes, built by the ray-MLP chain from the same rays)
    parity = {"sample_rays": rs, "sample": f"every {step_r}-th ray of the scene", "image": 0, "checker": "oracle/sixdgs_oracle.c (restates the reference; pinned by tests/golden g1..g13)"}
    try:
        parity["ray_mlp_keys"] = {"rays": rm, "max_row_rel_err": key_err,
                                  "note": "HIP ray MLP + k_proj (fp32 keys of the sample) against the oracle's own ray MLP + k_proj on the first rays of the sample, row-wise"}
    except NameError:
        pass
    try:
        gt0 = gts[0].cpu().numpy()
        up_dev = sol["up"][:1].contiguous()
        i2, v2, sc2 = idm.score_tokens([tok_dev], o_s, d_s, c_s, 100, want_scores=True)               # two-pass scorer: full score vector
        sc2 = sc2[0].cpu().numpy()
        parity["two_pass"] = {"score_rel_err": float(np.abs(sc2 - s).max() / s.max()),
                              "top100_identical": bool(set(i2[0].tolist()) == set(idx.tolist())),
                              "top100_same_order": bool(i2[0].tolist() == idx.tolist()),
                              "value_rel_err": float(np.abs(v2[0].cpu().numpy() - val).max() / val.max())}
        i1, v1, _ = idm.score_tokens([tok_dev], o_s, d_s, c_s, 100, want_scores=False)                 # the path of the timed steps
        parity["select"] = {"path": getattr(idm, "last_scoring_path", "?"),
                            "top100_identical": bool(set(i1[0].tolist()) == set(idx.tolist())),
                            "top100_same_order": bool(i1[0].tolist() == idx.tolist()),
                            "value_rel_err": float(np.abs(v1[0].cpu().numpy() - val).max() / val.max())}
        solp = ops.solve_pose(o_s, d_s, i1, v1, up_dev, gts[:1].contiguous())
        c_hip = solp["c2w"][0].cpu().numpy()
        rot, tr = _pose_delta(c_hip, p_ref["c2w"])
        e_hip, e_ref = _pose_delta(c_hip, gt0), _pose_delta(p_ref["c2w"], gt0)
        parity["pose"] = {"rot_err_deg": rot, "trans_err": tr, "c2w_max_abs_diff": float(np.abs(c_hip - p_ref["c2w"]).max()),
                          "vs_synthetic_gt": {"hip": {"rot_err_deg": e_hip[0], "trans_err": e_hip[1]},
                                              "oracle": {"rot_err_deg": e_ref[0], "trans_err": e_ref[1]},
                                              "kernel_reported": {"trans_err": float(solp["errors"][0, 0]), "rot_err_deg": float(solp["errors"][0, 1])}}}
        parity["score_rel_err"] = parity["two_pass"]["score_rel_err"]
        parity["top100_identical"] = bool(parity["two_pass"]["top100_identical"] and parity["select"]["top100_identical"])
        parity["rot_err_deg"], parity["trans_err"] = rot, tr
    except Exception as e:  # never lose the bench line to the checker
        parity["error"] = f"{e.__class__.__name__}: {e}"
        parity.setdefault("pose", {"rot_err_deg": None, "trans_err": None, "vs_synthetic_gt": None})
    # The port keeps the reference's scalar summation order (it is a parity oracle, not a tuned CPU code).  For scale, the
    # same per-pose path written with PyTorch CPU ops (what the reference runs: blocked multi-threaded GEMM + softmax + topk)
    # on the same sample, all host threads -- SURVEY 8(d) "the build's own PyTorch re-expression on device=cpu".
    try:
        torch.set_num_threads(max(1, os.cpu_count() or 1))
        k_t, tok_t = torch.from_numpy(key), torch.from_numpy(tok)
        wq, bq = torch.from_numpy(sd["attention.q_proj.weight"]), torch.from_numpy(sd["attention.q_proj.bias"])
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            qt = tok_t @ wq.t() + bq
            att = torch.softmax((qt @ k_t.t()) / (qt.shape[-1] ** 0.5), dim=-1)
            sc = att.sum(dim=0)
            torch.topk(sc, 100)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out["torch_cpu"] = {"value": round(1.0 / (best * (R / rs)), 6), "unit": "poses/s", "threads": torch.get_num_threads(),
                            "sample_seconds": round(best, 3), "note": "q_proj + softmax(QK^T) + column sum + top-100 with PyTorch CPU ops"}
    except Exception as e:  # the port above is the contract; this figure is informative
        out["torch_cpu"] = {"error": e.__class__.__name__}
    return out, parity


if __name__ == "__main__":
    main()
