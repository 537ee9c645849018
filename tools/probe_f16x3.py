import ctypes, os, numpy as np, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "libprobe6.so"))
torch.manual_seed(0)
def p(t): return ctypes.c_void_p(t.data_ptr())
def run6(A, B, mode):
    D = torch.zeros(32, 32, device="cuda")
    lib.probe_run(p(A), p(B), p(D), A.shape[1], mode, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)); torch.cuda.synchronize()
    return D.cpu().double()
def pow2scale(t, target=2.0 ** 14):
    m = float(t.abs().max())
    return float(2.0 ** np.floor(np.log2(target / m))) if m > 0 else 1.0
def runf(A, B, terms):
    D = torch.zeros(32, 32, device="cuda")
    lib.probe_run_f16(p(A), p(B), p(D), A.shape[1], ctypes.c_float(pow2scale(A)), ctypes.c_float(pow2scale(B)), terms,
                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)); torch.cuda.synchronize()
    return D.cpu().double()
gens = (("N(0,1)", lambda s: torch.randn(s)),
        ("rows scaled 1e-3..1e3", lambda s: torch.randn(s) * torch.logspace(-3, 3, s[0])[:, None]),
        ("rows scaled 1e-6..1e6", lambda s: torch.randn(s) * torch.logspace(-6, 6, s[0])[:, None]),
        ("elements lognormal sigma 4", lambda s: torch.randn(s) * torch.exp(4 * torch.randn(s))),
        ("positive U(0,1)", lambda s: torch.rand(s)), ("tiny 1e-20", lambda s: torch.randn(s) * 1e-20), ("huge 1e15", lambda s: torch.randn(s) * 1e15))
for name, gen in gens:
    st = {k: [] for k in ("f32 mfma", "bf16x6", "f16x3", "f16x4")}
    absmax = {k: [] for k in st}
    for trial in range(20):
        A = gen((32, 384)).cuda().contiguous(); B = gen((32, 384)).cuda().contiguous()
        ref = A.cpu().double() @ B.cpu().double().T
        den = (A.cpu().double().abs() @ B.cpu().double().abs().T)
        big = float(A.abs().max()) * float(B.abs().max()) * 384
        for k, D in (("f32 mfma", run6(A, B, 0)), ("bf16x6", run6(A, B, 6)), ("f16x3", runf(A, B, 3)), ("f16x4", runf(A, B, 4))):
            st[k].append(((D - ref).abs() / den).max().item()); absmax[k].append(((D - ref).abs().max() / big).item())
    print(f"{name:28s}", {k: f"{np.mean(v):.2e}/{np.max(v):.2e}" for k, v in st.items()}, " abs/(max|a|max|b|K):", {k: f"{np.max(v):.1e}" for k, v in absmax.items()})
