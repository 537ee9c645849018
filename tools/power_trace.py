"""Socket power / shader clock sampled at >= 10 Hz while ONE kernel family runs in a loop (VERDICT r3 #4a: "power-bound" as a measurement).

    python tools/power_trace.py <workload> [seconds]      workload = idle | sweep | chain | mfma | copy
      sweep : sixdgs_select_sweep (k_logits_f16x<kOutUB>), 4 images x 32 M rays of random key planes
      chain : sixdgs_ray_keys_ex (encode + five k_dense_planes launches per chunk), 8 M rays
      mfma  : the sweep of an -DSIXDGS_ABLATION build with everything but its MFMAs compiled out (= abl59)
      abl<N>: the sweep under compile-time ablation N (0 full, 2 no epilogue, 18 no fragment reads + no epilogue, 11 no DMA + no epilogue,
              27 MFMA + barriers, 59 MFMA only; names in tools/ablate_logits.py)
              (SIXDGS_LIB must point at build/variants/lib_abl.so: python tools/build_variant.py abl -DSIXDGS_ABLATION)
      copy  : torch copy of 8 GB (HBM only, no matrix pipe)

The sampler is a separate PROCESS (amdsmi python bindings; sysfs hwmon as a fallback), so that the launch loop's host thread does not
disturb the sampling period.  Prints one markdown table row + the raw samples into gpurun_out/power/<workload>.csv."""
import importlib, json, multiprocessing as mp, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _sysfs_paths():
    import glob
    hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
    return hw[0] if hw else None


def sampler(stop, path, period):
    """Writes `t, power_W, sclk_MHz, [per-XCD clocks], temp_C, throttle` rows until `stop` is set."""
    src = None
    h = None
    try:
        import amdsmi
        amdsmi.amdsmi_init()
        h = amdsmi.amdsmi_get_processor_handles()[0]
        src = "amdsmi"
    except Exception as e:                               # pragma: no cover
        src = None
        err = repr(e)
    hw = _sysfs_paths()
    meta = {"source": src or ("sysfs" if hw else "none")}
    if src == "amdsmi":
        for name, fn in (("power_cap", lambda: amdsmi.amdsmi_get_power_cap_info(h)), ("power_info", lambda: amdsmi.amdsmi_get_power_info(h)),
                         ("asic", lambda: amdsmi.amdsmi_get_gpu_asic_info(h))):
            try:
                meta[name] = {k: (v if isinstance(v, (int, float, str)) else str(v)) for k, v in fn().items()}
            except Exception as e:
                meta[name] = "n/a: " + repr(e)[:120]
    with open(path + ".meta.json", "w") as f:
        json.dump(meta, f)
    t0 = time.time()
    with open(path, "w") as f:
        f.write("t_s,power_w,sclk_mhz,xcd_clks,temp_c,throttle,mclk_mhz\n")
        while not stop.is_set():
            t = time.time() - t0
            p = clk = temp = thr = mclk = ""
            xcd = ""
            if src == "amdsmi":
                try:
                    m = amdsmi.amdsmi_get_gpu_metrics_info(h)
                    p = m.get("current_socket_power", m.get("average_socket_power", ""))
                    clk = m.get("current_gfxclk", m.get("average_gfxclk_frequency", ""))
                    g = m.get("current_gfxclks")
                    if isinstance(g, (list, tuple)):
                        g = [x for x in g if isinstance(x, (int, float)) and 0 < x < 10000]
                        xcd = "|".join(str(x) for x in g)
                        if g:
                            clk = sum(g) / len(g)
                    temp = m.get("temperature_hotspot", "")
                    thr = m.get("throttle_status", m.get("indep_throttle_status", ""))
                    mclk = m.get("current_uclk", "")
                except Exception:
                    try:
                        p = amdsmi.amdsmi_get_power_info(h).get("current_socket_power", "")
                        clk = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX).get("clk", "")
                    except Exception:
                        pass
            elif hw:
                try:
                    for n in ("power1_input", "power1_average"):
                        fp = os.path.join(hw, n)
                        if os.path.exists(fp):
                            p = int(open(fp).read()) / 1e6
                            break
                    fp = os.path.join(hw, "freq1_input")
                    if os.path.exists(fp):
                        clk = int(open(fp).read()) / 1e6
                except Exception:
                    pass
            f.write(f"{t:.3f},{p},{clk},{xcd},{temp},{thr},{mclk}\n")
            f.flush()
            dt = period - ((time.time() - t0) % period)
            time.sleep(dt)


def main():
    wl = sys.argv[1]
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
    out = os.path.join(ROOT, "gpurun_out", "power")
    os.makedirs(out, exist_ok=True)
    csv = os.path.join(out, wl + ".csv")
    ctx = mp.get_context("spawn")
    stop = ctx.Event()
    proc = ctx.Process(target=sampler, args=(stop, csv, 0.05))
    proc.start()
    import torch
    os.environ.setdefault("SIXDGS_RANDOM_BACKBONE", "1")
    note = ""
    if wl == "idle":
        torch.zeros(1, device="cuda")
        time.sleep(secs)
        launches, per_ms = 0, 0.0
        t_lo, t_hi = 0.5, secs
    else:
        ops = importlib.import_module("6dgs_amd.ops")
        syn = importlib.import_module("6dgs_amd.synthetic")
        torch.manual_seed(0)
        abl = 59 if wl == "mfma" else (int(wl[3:]) if wl.startswith("abl") else None)
        if wl == "sweep" or abl is not None:
            B, R = 4, int(os.environ.get("POWER_RAYS", 32_000_000))
            planes = torch.empty(R, 1536, dtype=torch.uint8, device="cuda")
            scale = torch.empty((R + 127) // 128, device="cuda")
            for r0 in range(0, R, 1 << 20):
                k = torch.randn(min(1 << 20, R - r0), 384, device="cuda") * 0.07
                p, s = ops.split_planes_f16(k)
                planes[r0:r0 + k.shape[0]] = p
                scale[r0 // 128:r0 // 128 + s.shape[0]] = s
            q = torch.randn(B, 256, 384, device="cuda")
            nt = torch.full((B,), 256, dtype=torch.int32, device="cuda")
            si = ops.select_sample_indices(min(R, 1 << 22), "cuda")
            sp = planes[si].contiguous()
            _, ssc = ops.split_planes_f16(torch.randn(si.shape[0], 384, device="cuda") * 0.07)
            ss = ops.SelectStream(q, nt, R, 100, 4096, [256] * B)
            ss.begin(sp, ssc)
            fl = 2.0 * 256 * 384 * R * B
            if abl is not None:
                os.environ["SIXDGS_DEBUG_ABLATE"] = str(abl)
                note = f"select sweep k_logits_f16x<UB> under compile-time ablation {abl} (2 no epilogue, 18 no fragment reads + no epilogue, 11 no DMA + no epilogue, 27 MFMA + barriers, 59 MFMA only, 4096 idle-wave skip off)"

            def one():
                ss.sweep(planes, scale, 0, None, update_norm=False)
        elif wl == "chain":
            R = 8 << 20
            rays = syn.make_rays(1 << 20, 0)
            o, d, c = (torch.from_numpy(rays[k]).cuda().repeat(R >> 20, 1).contiguous() for k in ("ori", "dir", "rgb"))
            w = ops.PackedWeights({k: torch.from_numpy(v) for k, v in syn.make_scorer_state_dict(0).items()}, "cuda")
            fl = 2025472.0 * R
            keep = {}

            def one():
                keep["o"] = ops.ray_keys(o, d, c, w, want_key=False, want_planes=True)
        elif wl == "copy":
            a = torch.empty(8 << 30, dtype=torch.uint8, device="cuda")
            b = torch.empty_like(a)
            fl = 0.0

            def one():
                b.copy_(a)
        else:
            raise SystemExit("unknown workload " + wl)
        one(); torch.cuda.synchronize()
        time.sleep(1.0)
        t_lo = time.time()
        ev = []
        launches = 0
        t_start = time.time()
        while time.time() - t_start < secs:
            a0, b0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record(); one(); b0.record(); b0.synchronize()
            ev.append(a0.elapsed_time(b0)); launches += 1
        t_hi = time.time()
        per_ms = sorted(ev)[len(ev) // 2]
        first, last = sum(ev[:3]) / 3, sum(ev[-3:]) / 3
        note += f" first3 {first:.2f} ms last3 {last:.2f} ms"
    time.sleep(0.5)
    stop.set(); proc.join(5)
    # ---- summary over the samples that fall inside the loop (the sampler's clock starts at its own t0: align by wall time of file creation)
    meta = json.load(open(csv + ".meta.json"))
    rows = [l.strip().split(",") for l in open(csv).read().splitlines()[1:]]
    t_file0 = os.path.getctime(csv)
    def num(x):
        try:
            return float(x)
        except Exception:
            return None
    if wl == "idle":
        sel = rows
    else:
        t_csv0 = None
        # sampler wrote t relative to its own start; it started before the workload, so use the LAST (t_hi - t_lo) seconds minus the 0.5 s tail
        tmax = num(rows[-1][0])
        sel = [r for r in rows if tmax - 0.5 - (t_hi - t_lo) + 1.0 <= num(r[0]) <= tmax - 0.7]
    pw = [num(r[1]) for r in sel if num(r[1]) is not None]
    ck = [num(r[2]) for r in sel if num(r[2]) is not None]
    tp = [num(r[4]) for r in sel if num(r[4]) is not None]
    mk = [num(r[6]) for r in sel if len(r) > 6 and num(r[6]) is not None]
    thr = sorted({r[5] for r in sel})
    def stat(v):
        if not v:
            return "n/a"
        v = sorted(v)
        return f"{sum(v) / len(v):.0f} (min {v[0]:.0f}, p50 {v[len(v) // 2]:.0f}, max {v[-1]:.0f})"
    dt = [num(rows[i + 1][0]) - num(rows[i][0]) for i in range(len(rows) - 1)]
    hz = 1.0 / (sum(dt) / len(dt)) if dt else 0.0
    cap = meta.get("power_cap")
    cap_w = None
    if isinstance(cap, dict):
        cap_w = cap.get("power_cap")
        if isinstance(cap_w, (int, float)) and cap_w > 10000:
            cap_w = cap_w / 1e6
    res = dict(workload=wl, seconds=secs, launches=launches, median_launch_ms=round(per_ms, 3), tflops_fp32_eq=(round(fl / per_ms / 1e9, 1) if wl != "idle" and per_ms and fl else None),
               samples=len(sel), sample_hz=round(hz, 1), power_w=stat(pw), sclk_mhz=stat(ck), mclk_mhz=stat(mk), temp_hotspot_c=stat(tp), throttle=thr[:6], power_cap_w=cap_w,
               source=meta.get("source"), note=note.strip())
    print("POWER", json.dumps(res))
    with open(os.path.join(out, wl + ".json"), "w") as f:
        json.dump(dict(res, meta=meta), f, indent=1)


if __name__ == "__main__":
    main()
