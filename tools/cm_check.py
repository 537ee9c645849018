"""Key planes of the plane-to-plane chain under the activation layout of THIS process (SIXDGS_DENSE_CM=0/1): `save <file>` writes them,
`compare <file>` checks bit-equality against a saved run of the other layout and prints where any difference sits.

    SIXDGS_DENSE_CM=0 python tools/cm_check.py save /tmp/cm0.pt; SIXDGS_DENSE_CM=1 python tools/cm_check.py compare /tmp/cm0.pt

"""
import importlib, os, sys
os.environ.setdefault("SIXDGS_RANDOM_BACKBONE", "1")
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
syn = importlib.import_module("6dgs_amd.synthetic"); ops = importlib.import_module("6dgs_amd.ops")
w = ops.PackedWeights({k: torch.from_numpy(v) for k, v in syn.make_scorer_state_dict(11).items()}, "cuda")
out = {}
for R, chunk in ((128 * 301 + 77, 262144), (128 * 301 + 77, 128 * 90), (5, 262144), (300, 262144), (256 * 300 - 3, 262144)):
    rays = syn.make_rays(R, 31)
    rays["ori"] = (rays["ori"] * np.logspace(-2, 3, R)[:, None]).astype(np.float32)
    o, d, c = (torch.from_numpy(rays[k]).cuda() for k in ("ori", "dir", "rgb"))
    norm = torch.zeros(1, device="cuda")
    _, _, (planes, inv) = ops.ray_keys(o, d, c, w, want_key=False, want_planes=True, max_chunk=chunk, norm_out=norm)
    _, key, _ = ops.ray_keys(o, d, c, w, want_key=True, want_planes=True, max_chunk=chunk)
    out[f"{R}_{chunk}"] = dict(planes=planes.cpu(), inv=inv.cpu(), key=key.cpu(), norm=norm.cpu())
mode = {"0": "ray-major", "1": "chunk-major"}.get(os.environ.get("SIXDGS_DENSE_CM", ""), "default (chunk-major)")
if sys.argv[1] == "save":
    torch.save(out, sys.argv[2])
    print(f"saved {len(out)} cases, layout {mode}")
else:
    ref = torch.load(sys.argv[2])
    bad = 0
    for k, v in out.items():
        r = ref[k]
        R = v["key"].shape[0]
        same = torch.equal(v["planes"], r["planes"]) and torch.equal(v["inv"], r["inv"]) and torch.equal(v["key"], r["key"]) and torch.equal(v["norm"], r["norm"])
        print(f"case {k}: {'identical' if same else 'DIFFERENT'} (layout {mode} vs saved)")
        if not same:
            bad += 1
            dk = (v["key"] != r["key"])
            print("  fp32 keys: differing elements", int(dk.sum()), "of", dk.numel(), "; rays with a difference", int(dk.any(1).sum()), "; nan", int(torch.isnan(v["key"]).sum()))
            print("  by feature block of 32:", dk.float().mean(0).view(12, 32).mean(1).numpy().round(3).tolist())
            print("  by feature mod 32:", dk.float().mean(0).view(12, 32).mean(0).numpy().round(2).tolist())
            print("  by ray mod 256 (buckets of 32):", torch.stack([dk[i::256].float().mean() for i in range(0, 256, 32)]).numpy().round(3).tolist())
            print("  by ray position (8 equal parts):", [round(float(x.float().mean()), 3) for x in dk.any(1).chunk(8)])
            rel = ((v["key"] - r["key"]).abs().max() / r["key"].abs().max()).item()
            print("  max |diff| / max |key|:", rel, " inv equal:", torch.equal(v["inv"], r["inv"]), " norm:", float(v["norm"]), float(r["norm"]))
    print("CM_CHECK", "FAIL" if bad else "PASS")
    sys.exit(1 if bad else 0)
