"""BASELINE.json configs[4] -- "full Mip-NeRF360 + Tanks&Temples eval sweep (all scenes, all test views)" -- at the WORKLOAD's scale with
stand-ins for what cannot exist offline (VERDICT r3 #5b).  `python bench.py --config cfg5-standin` runs it.

What is real: the twelve scenes the reference sweeps (tools/launch_all_mip_training.sh:3-9, tools/launch_all_tanks_and_temple_training.sh:3-7;
pretrain_eval_attention.py:200-248 walks them one after the other), each at the Gaussian count 3DGS reports for it after 30 k iterations
(Mip-NeRF360) or at the size of a masked single-object reconstruction (Tanks&Temples / NSVF; BASELINE.json: Ignatius ~300 k), with the
reference's number of TEST views (every 8th image of the COLMAP scenes, scene/colmap.py; the `2_*` split of the NSVF scenes,
scene/tanksandtemples.py), 800-pixel queries, RGBA queries with an object mask for the T&T scenes (the wrapper then keeps 50-150 of the 256
tokens, backbone.py:86-114), rays from EVERY valid Gaussian x 64 iso-cell rays (north_star).  Scenes whose key planes (1536 B per ray) fit the
GPU are scored resident, the others (bicycle, garden, stump: 317-392 M rays) streamed.
What is synthetic: Gaussian parameters, pixels, scorer / backbone weights (random init) -- poses are meaningless, throughput and parity are not.
Per scene ONE view is checked against the CPU oracle on a prefix of the scene's rays (the oracle's OWN ray MLP + k_proj on those rays, its
softmax scorer, top-100 and pose solve against the HIP path on the same prefix and the same tokens): the whole scene would cost the
oracle 6-9 minutes per view.

Multi-rank: like the evaluation sweep every rank builds every scene and takes a contiguous block of its test views."""
import time

import numpy as np

#        name                 Gaussians  test views  masked
SCENES = [
    ("mip_360_bicycle", 6_130_000, 25, False),
    ("mip_360_bonsai", 1_240_000, 37, False),
    ("mip_360_counter", 1_220_000, 30, False),
    ("mip_360_garden", 5_830_000, 24, False),
    ("mip_360_kitchen", 1_850_000, 35, False),
    ("mip_360_room", 1_590_000, 39, False),
    ("mip_360_stump", 4_960_000, 16, False),
    ("tt_Barn", 600_000, 48, True),
    ("tt_Caterpillar", 500_000, 46, True),
    ("tt_Family", 350_000, 19, True),
    ("tt_Ignatius", 300_000, 33, True),
    ("tt_Truck", 450_000, 32, True),
]


def masked_views(syn, n, seed, size):
    """RGBA queries whose alpha is an object silhouette (an off-centre ellipse, a third to a half of the frame): the mask -> token selection of
    the wrapper keeps the tokens under it."""
    cams = syn.make_cameras(n, seed, width=size, height=size, rgba=True)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    rng = np.random.default_rng(seed)
    for cam in cams:
        cx, cy = size * (0.5 + 0.08 * rng.standard_normal()), size * (0.5 + 0.08 * rng.standard_normal())
        ax, ay = size * rng.uniform(0.24, 0.38), size * rng.uniform(0.24, 0.38)
        a = (((xx - cx) / ax) ** 2 + ((yy - cy) / ay) ** 2) <= 1.0
        img = cam["image"].copy()
        img[..., 3] = np.where(a, 255, 0).astype(np.uint8)
        cam["image"] = img
    return cams


def run(args, torch, pkg, syn, dd, ops, tp, idm, dev, rank, world, peak_tflops):
    from oracle import oracle as O              # the checker (bench.py's parity leg only; the timed path never touches it)
    O.build()
    scenes = [s for s in SCENES if not args.scenes or any(k in s[0] for k in args.scenes.split(","))]
    if args.scale != 1.0:
        scenes = [(n, max(20_000, int(g * args.scale)), v, m) for n, g, v, m in scenes]
    sd = {k: v.detach().cpu().numpy() for k, v in idm._scorer_params().items()}
    hbm = torch.cuda.get_device_properties(dev).total_memory
    rows, tot_views, tot_eval, tot_setup, tot_standin = [], 0, 0.0, 0.0, 0.0
    # ONE arena for the big per-scene buffers (key planes, select workspace, chain workspace, a streamed scene's U and chunk planes): allocated once, carved
    # per scene, no empty_cache() and no 100-200 GB hipMalloc between scenes (VERDICT r4 #3b; ops.Arena).  --no-arena: PyTorch's caching allocator as in round 4.
    arena_on = not getattr(args, "no_arena", False)
    t_arena = 0.0
    if arena_on:
        torch.cuda.synchronize()
        t_a0 = time.perf_counter()
        free_b = torch.cuda.mem_get_info(dev)[0]
        arena = ops.Arena(max(1 << 30, int(free_b - (56 << 30))) if free_b > (80 << 30) else int(0.6 * free_b), dev)     # rays, images, tokens and torch's own needs stay outside
        arena.buf[:: 1 << 21].zero_()
        torch.cuda.synchronize()
        t_arena = time.perf_counter() - t_a0
        ops.set_arena(arena)
    sweep_ms, sweep_fl, sweep_n = 0.0, 0.0, 0
    chain_ms, chain_fl = 0.0, 0.0
    cpu_sample = None
    for si, (name, n_gauss, n_views, masked) in enumerate(scenes):
        if args.views_cap:
            n_views = min(n_views, args.views_cap)
        # ---- scene set-up, itemised (VERDICT r4 #3a): every item ends in a device sync, the items add up to setup_s.  "standin" items exist only because
        # the scene and its views are synthesised here (host RNG over 59 floats per Gaussian, camera / silhouette synthesis); the reference reads a PLY and
        # a camera file instead (pretrain_eval_attention.py:89,107,136).  The rest is what the product pays once per scene.
        t_s0 = time.perf_counter()
        items, t_last = {}, [t_s0]

        def lap(name):
            torch.cuda.synchronize()
            now = time.perf_counter()
            items[name] = round(items.get(name, 0.0) + now - t_last[0], 3)
            t_last[0] = now

        host_scene = syn.make_scene(n_gauss, 100 + si) if rank == 0 else None
        lap("standin_host_scene_generation")
        scene = pkg.GaussianScene.from_dict(host_scene, device=dev) if rank == 0 else None
        del host_scene
        scene = dd.broadcast_scene(scene, 0, device=dev)
        lap("scene_upload_and_broadcast")
        ori, dr, rgb = pkg.generate_all_possible_rays(scene, max_ellipsoids=-1, emitter="isocell", rays_per_ellipsoid=args.rays_per_ellipsoid)
        fin = torch.isfinite(dr).all(dim=1)
        if not bool(fin.all()):
            ori, dr, rgb = ori[fin].contiguous(), dr[fin].contiguous(), rgb[fin].contiguous()
        del fin, scene
        R = int(ori.shape[0])
        lap("normals_knn_and_emission")
        idm.invalidate_caches()
        if not arena_on:
            torch.cuda.empty_cache()
        lap("release_previous_scene")
        budget = arena.capacity if arena_on else 0.85 * hbm
        resident = R * 1536 + R // 8 * 96 + ops.ray_keys_workspace_bytes(R, ops.RAY_KEYS_CHUNK_MIN) + 24 * R * min(getattr(args, "masked_batch", args.batch) if masked else args.batch, max(n_views, 1)) < 0.97 * budget
        if getattr(args, "stream_above_rays", 0) and R > args.stream_above_rays:
            resident = False
        kprof = ops.KernelProfile()
        if resident:
            if not arena_on:       # the plane buffer's first hipMalloc (pages mapped and cleared at 30-50 GB/s), timed on its own; the allocator hands it to _ensure_keys
                tmp = torch.empty(R, 1536, dtype=torch.uint8, device=dev)
                lap("key_plane_buffer_alloc")
                del tmp
            idm._ensure_keys(ori, dr, rgb, profile=kprof)
        lap("ray_mlp_key_planes_and_sample")
        k_ms, k_fl, _, _ = kprof.collect()
        chain_ms, chain_fl = chain_ms + k_ms, chain_fl + k_fl
        lo, hi = dd.shard_range(n_views, rank, world)
        cams = (masked_views(syn, n_views, 700 + si, args.image_size) if masked else syn.make_cameras(n_views, 700 + si, width=args.image_size, height=args.image_size))[lo:hi]
        lap("standin_host_view_synthesis")
        images = [torch.from_numpy(c["image"]).to(dev) for c in cams]
        gts = torch.stack([tp.gt_pose_and_intrinsics(pkg.CameraInfo(**c), dev)[0] for c in cams]).to(dev) if cams else None
        lap("views_upload")
        batch = max(1, min(len(images), (getattr(args, "masked_batch", args.batch) if masked else args.batch) if resident else args.streamed_batch))
        if images:                                   # batches of equal size (19 views: 10 + 9, not 16 + 3): a short last batch sweeps at a near-empty launch's efficiency
            batch = -(-len(images) // (-(-len(images) // batch)))
        if not resident and images and not arena_on:
            # A streamed scene's buffers -- U (4 B per ray and image), one chunk's key planes and chain workspace -- are allocated inside the step.  Their first
            # hipMalloc is scene set-up like the resident scenes' plane buffer, and it is NOT cheap behind a scene that just returned 200 GB to the driver: the
            # pages are cleared on the way back out (measured: the same streamed step 5.1 s on its own, 8.1 s behind a resident scene).  Touch them here once.
            n_c = min(args.chunk_rays, R)
            warm = [torch.empty(batch * ((R + 255) // 256 * 256) * 4, dtype=torch.uint8, device=dev), torch.empty(n_c * 1536, dtype=torch.uint8, device=dev),
                    torch.empty(ops.ray_keys_workspace_bytes(n_c), dtype=torch.uint8, device=dev)]
            for w_ in warm:
                w_[:: 1 << 20].zero_()
            del warm, w_
            lap("streamed_buffers_first_touch")
        if resident and images:      # scene set-up, like the key planes: the first batch once untimed, so that the select workspace (20 B per ray and image:
            tp.estimate_poses(idm, images[:batch], ori, dr, rgb, gt_c2w=gts[:batch], defer_status=True)      # a 10-40 GB hipMalloc at ~50 GB/s) and the image-side graph exist
            lap("first_batch_untimed_select_workspace_and_image_graph")
        torch.cuda.synchronize()
        dd.barrier()
        t_setup = time.perf_counter() - t_s0
        lap("barrier")
        # ---- the timed part: all test views of the scene, batch by batch, poses on the host at the end of every batch
        prof = ops.KernelProfile()
        tok_counts, first = [], None
        mem0 = (round(torch.cuda.memory_allocated(dev) / 2**30, 1), round(torch.cuda.memory_reserved(dev) / 2**30, 1), round(torch.cuda.mem_get_info(dev)[0] / 2**30, 1))
        step_s = []
        t0 = time.perf_counter()
        def note(sol, c2w):
            nonlocal first
            tk = sol["tokens"]                                   # a list (ragged: masked views), a dense block or a BatchedTokens: all index per image
            tok_counts.extend(int(tk[i].shape[0]) for i in range(len(tk)))
            if first is None:
                first = {"tokens": tk[0].contiguous().clone(), "up": sol["up"][:1].clone(), "c2w": c2w[0].clone()}

        if resident and not getattr(args, "no_pipeline", False):
            # resident scenes: two batches in flight (tp.PoseStream), as the evaluation loop of test_pose_estimation runs them; step_s = intervals between completions
            ps, prev, t_b = tp.PoseStream(idm, ori, dr, rgb), None, time.perf_counter()
            for b0 in list(range(0, len(images), batch)) + [None]:
                cur = ps.submit(images[b0:b0 + batch], gts[b0:b0 + batch], profile=prof) if b0 is not None else None
                if prev is not None:
                    c2w, sol = ps.collect(prev)
                    note(sol, c2w)
                    step_s.append(round(time.perf_counter() - t_b, 3))
                    t_b = time.perf_counter()
                    del sol
                prev = cur
        else:
            for b0 in range(0, len(images), batch):
                t_b = time.perf_counter()
                sol = tp.estimate_poses(idm, images[b0:b0 + batch], ori, dr, rgb, gt_c2w=gts[b0:b0 + batch], profile=prof,
                                        streamed_chunk_rays=None if resident else args.chunk_rays, defer_status=resident)
                if "packed" in sol:
                    c2w = tp.resolve_poses(idm, sol, sol["packed"].cpu())
                else:
                    c2w = sol["c2w"].cpu()
                note(sol, c2w)
                step_s.append(round(time.perf_counter() - t_b, 3))
                del sol
        torch.cuda.synchronize()
        dd.barrier()
        t_eval = dd.max_over_ranks(time.perf_counter() - t0, dev)
        l_ms, l_fl, _, l_n = prof.collect()
        sweep_ms, sweep_fl, sweep_n = sweep_ms + l_ms, sweep_fl + l_fl, sweep_n + l_n
        path = getattr(idm, "last_scoring_path", "?")
        row = {"scene": name, "gaussians": n_gauss, "rays": R, "test_views": n_views, "masked": masked, "scoring": "resident" if resident else "streamed",
               "scoring_path": path, "images_per_step": batch, "tokens_per_image_mean": round(float(np.mean(tok_counts)), 1) if tok_counts else None,
               "tokens_per_image_min_max": [int(min(tok_counts)), int(max(tok_counts))] if tok_counts else None,
               "setup_s": round(t_setup, 2), "setup_breakdown_s": items,
               "setup_standin_s": round(sum(v for k_, v in items.items() if k_.startswith("standin_")), 3), "ray_mlp_keys_tflops": round(k_fl / (k_ms * 1e-3) / 1e12, 1) if k_ms > 0 else None,
               "eval_s": round(t_eval, 3), "poses_per_s": round(n_views / t_eval, 3), "step_s": step_s,
               "gib_allocated_reserved_free_before_eval": mem0,
               "sweep_tflops": round(l_fl / (l_ms * 1e-3) / 1e12, 1) if l_ms > 0 else None}
        # ---- one view of the scene against the oracle, on a prefix of its rays (rank 0)
        if rank == 0 and first is not None and not args.skip_cpu_baseline:
            rs = int(min(args.oracle_rays, R))
            o_s, d_s, c_s = ori[:rs].contiguous(), dr[:rs].contiguous(), rgb[:rs].contiguous()
            tok = first["tokens"]
            i1, v1, _ = idm.score_tokens([tok], o_s, d_s, c_s, 100, want_scores=False)
            p1 = getattr(idm, "last_scoring_path", "?")
            i2, v2, sc2 = idm.score_tokens([tok], o_s, d_s, c_s, 100, want_scores=True)
            solp = ops.solve_pose(o_s, d_s, i1, v1, first["up"])
            t1 = time.perf_counter()
            _, okey = O.ray_features(o_s.cpu().numpy(), d_s.cpu().numpy(), c_s.cpu().numpy(), sd, want_feat=False)
            t_mlp = time.perf_counter() - t1
            t1 = time.perf_counter()
            s = O.attention_scores(O.q_proj(tok.cpu().numpy(), sd), okey)
            oi, ov = O.topk(s, 100)
            p_ref = O.pose_from_topk(o_s.cpu().numpy(), d_s.cpu().numpy(), oi, ov, first["up"][0].cpu().numpy())
            t_pose = time.perf_counter() - t1
            scale = max(1.0, float(np.abs(p_ref["c2w"]).max()))
            row["parity_vs_oracle"] = {
                "sample_rays": rs, "tokens": int(tok.shape[0]), "select_path": p1,
                "score_rel_err": float(np.abs(sc2[0].cpu().numpy() - s).max() / s.max()),
                "top100_identical_two_pass": bool(set(i2[0].tolist()) == set(oi.tolist())),
                "top100_identical_select": bool(set(i1[0].tolist()) == set(oi.tolist())),
                "value_rel_err_select": float(np.abs(v1[0].cpu().numpy() - s[i1[0].cpu().numpy()]).max() / s.max()),
                "pose_rel_err": float(np.abs(solp["c2w"][0].cpu().numpy() - p_ref["c2w"]).max() / scale),
                # the metric's own error measures (error_computation.py:3-8) between the HIP pose and the ORACLE's pose of the same view
                # (rotation angle from the chord |R_h - R_o|_F = 2 sqrt(2) sin(theta / 2): acos of the trace has a floor of ~0.02 deg for fp32 matrices)
                "rot_err_deg_vs_oracle": float(np.degrees(2.0 * np.arcsin(min(1.0, np.linalg.norm(solp["c2w"][0, :3, :3].cpu().numpy().astype(np.float64) - p_ref["c2w"][:3, :3].astype(np.float64)) / (2.0 * np.sqrt(2.0)))))),
                "trans_err_vs_oracle": float(np.linalg.norm(solp["c2w"][0, :3, 3].cpu().numpy().astype(np.float64) - p_ref["c2w"][:3, 3].astype(np.float64))),
                "oracle_s": {"ray_mlp": round(t_mlp, 2), "scorer_topk_pose": round(t_pose, 2)}}
            if cpu_sample is None:
                cpu_sample = (rs, R, t_mlp, t_pose, name)
            idm.invalidate_caches()
            del okey, s, sc2
        rows.append(row)
        tot_standin += row["setup_standin_s"]
        tot_views, tot_eval, tot_setup = tot_views + n_views, tot_eval + t_eval, tot_setup + t_setup
        del ori, dr, rgb, images, gts
        idm.invalidate_caches()
        if not arena_on:
            torch.cuda.empty_cache()  # (a 100-200 GB plane buffer left in the caching allocator gets split by small tensors and can then neither be reused for the next
                                      #  scene's planes nor released: out of memory on the fourth scene when this call was left out)
    ok = [r["parity_vs_oracle"] for r in rows if "parity_vs_oracle" in r]
    out = {
        "metric": "poses/sec", "value": round(tot_views / tot_eval, 4), "unit": "poses/s", "n_gpus": world, "steps": len(rows), "warmup": 0,
        "ms_per_step": round(1e3 * tot_eval / max(len(rows), 1), 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"cfg5-standin: the reference's 12-scene evaluation sweep at its scale -- {len(rows)} synthetic scenes at the Mip-NeRF360 / "
                               f"Tanks&Temples Gaussian counts ({min(r['gaussians'] for r in rows)}..{max(r['gaussians'] for r in rows)}), iso-cell emission from "
                               f"every valid Gaussian x {args.rays_per_ellipsoid} rays, {tot_views} test views in all ({args.image_size}x{args.image_size}, RGBA with an "
                               "object mask for the T&T scenes), 256 tokens x 384 or fewer, top-100; a 'step' is one scene",
                   "preset": "cfg5-standin", "scenes": len(rows), "test_views": tot_views, "rays_per_ellipsoid": args.rays_per_ellipsoid,
                   "parallelism": f"image-sharded x{world} (every rank builds every scene and takes a contiguous block of its test views)"},
        "value_including_scene_setup": round(tot_views / (tot_eval + tot_setup), 4),
        # set-up the PRODUCT pays per scene (upload, normals + emission, key planes, sample, workspaces) -- without the synthesis of the stand-in scene and views
        "value_including_product_scene_setup": round(tot_views / (tot_eval + tot_setup - tot_standin), 4),
        "process_setup_s": {"arena_alloc": round(t_arena, 2), "arena_gib": round(arena.capacity / 2**30, 1) if arena_on else 0.0,
                            "arena_high_water_gib": round(arena.high / 2**30, 1) if arena_on else 0.0,
                            "note": "one device buffer for the big per-scene buffers, allocated once per process (ops.Arena); not part of any scene's set-up"},
        "scene_setup_s_total": round(tot_setup, 2), "scene_setup_standin_s_total": round(tot_standin, 2), "eval_s_total": round(tot_eval, 2),
        "scene_setup_breakdown_s_total": {k_: round(sum(r["setup_breakdown_s"].get(k_, 0.0) for r in rows), 2) for k_ in sorted({k2 for r in rows for k2 in r["setup_breakdown_s"]})},
        "ranks_seen": dd.ranks_seen(dev), "backend": dd.backend_name(),
        "scenes": rows,
        "parity_summary": {"scenes_checked": len(ok),
                           "all_top100_identical": bool(ok) and all(p["top100_identical_two_pass"] and p["top100_identical_select"] for p in ok),
                           "max_score_rel_err": max((p["score_rel_err"] for p in ok), default=None),
                           "max_pose_rel_err": max((p["pose_rel_err"] for p in ok), default=None),
                           "mean_rot_err_deg_vs_oracle": float(np.mean([p["rot_err_deg_vs_oracle"] for p in ok])) if ok else None,
                           "mean_trans_err_vs_oracle": float(np.mean([p["trans_err_vs_oracle"] for p in ok])) if ok else None},
        "roofline": {"kernel": "k_logits_f16x<UB> (select sweep) over all scenes", "bound": "mfma", "achieved": round(sweep_fl / (sweep_ms * 1e-3) / 1e12, 2) if sweep_ms > 0 else None,
                     "peak": round(peak_tflops / 3, 1), "unit": "TFLOP/s", "frac": round(sweep_fl / (sweep_ms * 1e-3) / 1e12 / (peak_tflops / 3), 4) if sweep_ms > 0 else None,
                     "traffic": None, "launches": sweep_n,
                     "note": "algorithmic 2*T*384 FLOP per ray and image with T = the image's REAL token count (masked views credit only their tokens) / HIP-event time",
                     "ray_mlp_chain_tflops": round(chain_fl / (chain_ms * 1e-3) / 1e12, 1) if chain_ms > 0 else None},
    }
    if arena_on:
        ops.set_arena(None)
    if cpu_sample is not None:
        rs, R0, t_mlp, t_pose, nm = cpu_sample
        per_pose = (t_mlp + t_pose) * (R0 / rs)
        out["cpu_baseline"] = {"value": round(1.0 / per_pose, 6), "unit": "poses/s", "cores": O.num_threads(), "kind": "port",
                               "sample": f"scene {nm}: the oracle's ray MLP + k_proj ({t_mlp:.1f} s) and per-pose path ({t_pose:.1f} s) on the first {rs} of {R0} rays, one "
                                         "view, scaled by R/sample (the reference recomputes the ray MLP per image, identification_module.py:79)"}
    return out
