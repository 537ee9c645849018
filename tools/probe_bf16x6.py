import ctypes, os, numpy as np, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "libprobe6.so"))
torch.manual_seed(0)
def run(A, B, mode):
    D = torch.zeros(32, 32, device="cuda")
    rc = lib.probe_run(ctypes.c_void_p(A.data_ptr()), ctypes.c_void_p(B.data_ptr()), ctypes.c_void_p(D.data_ptr()), A.shape[1], mode,
                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize(); assert rc == 0
    return D.cpu().double()
for name, gen in (("N(0,1)", lambda s: torch.randn(s)), ("N(0,1)*[1e-3..1e3] rows", lambda s: torch.randn(s) * torch.logspace(-3, 3, s[0])[:, None]),
                  ("positive U(0,1)", lambda s: torch.rand(s)), ("tiny 1e-20", lambda s: torch.randn(s) * 1e-20)):
    stats = {m: [] for m in (0, 1, 3, 6, 7)}
    for trial in range(20):
        A = gen((32, 384)).cuda().contiguous(); B = gen((32, 384)).cuda().contiguous()
        ref = A.cpu().double() @ B.cpu().double().T
        den = (A.cpu().double().abs() @ B.cpu().double().abs().T)
        for m in stats:
            D = run(A, B, m)
            stats[m].append(((D - ref).abs() / den).max().item())
    print(name, {({0: "f32 mfma", 1: "bf16x1", 3: "bf16x3", 6: "bf16x6", 7: "bf16x6 sep-acc"}[m]): f"{np.mean(v):.2e}/{np.max(v):.2e}" for m, v in stats.items()})
# layout check with asymmetric operands: D must equal A B^T (not transposed), checked above via error vs ref
