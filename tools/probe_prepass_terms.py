"""Sample statistics of the select pre-pass with one MFMA term against three (SIXDGS_PREPASS_TERMS): run once per setting, dumps [B,256,2] to the given file."""
import importlib, os, sys
os.environ.setdefault("SIXDGS_RANDOM_BACKBONE", "1")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
syn = importlib.import_module("6dgs_amd.synthetic"); ops = importlib.import_module("6dgs_amd.ops")
R = 1 << 20
rays = syn.make_rays(R, 0)
o, d, c = (torch.from_numpy(rays[k]).cuda() for k in ("ori", "dir", "rgb"))
w = ops.PackedWeights({k: torch.from_numpy(v) for k, v in syn.make_scorer_state_dict(0).items()}, "cuda")
_, _, (planes, scale) = ops.ray_keys(o, d, c, w, want_key=False, want_planes=True)
g = torch.Generator().manual_seed(1)
q = (torch.randn(2, 256, 384, generator=g) * 1.5).cuda()
n_tok = torch.tensor([256, 200], dtype=torch.int32, device="cuda")
ss = ops.SelectStream(q, n_tok, R, n_tok_host=[256, 200])
st = ss.sample_stats(planes, scale)
torch.cuda.synchronize()
torch.save(st.cpu(), sys.argv[1])
print(sys.argv[1], st[0, :4].tolist(), bool(torch.isfinite(st[0]).all()), bool(torch.isfinite(st[1, :200]).all()))
