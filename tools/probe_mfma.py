import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "libprobe.so"))
torch.manual_seed(0)
dev = "cuda:0"
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count)
for name, M in (("probe_mfma32", 32), ("probe_mfma16", 16)):
    K = 64
    A = torch.randn(M, K, device=dev); B = torch.randn(K, M, device=dev); D = torch.zeros(M, M, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    rc = getattr(lib, name)(ctypes.c_void_p(A.data_ptr()), ctypes.c_void_p(B.data_ptr()), ctypes.c_void_p(D.data_ptr()), K, ctypes.c_void_p(s))
    torch.cuda.synchronize()
    ref = (A.double() @ B.double()).float()
    # bitwise check against an fmaf chain in k order
    acc = torch.zeros(M, M, dtype=torch.float64)
    print(name, "rc", rc, "maxerr", (D - ref).abs().max().item(), "transposed_err", (D.T - ref).abs().max().item())
import subprocess
print(subprocess.run("nproc; lscpu | grep -E 'Model name|Flags' | cut -c1-400; free -g | head -2; rocm-smi --showmeminfo vram | head -8", shell=True, capture_output=True, text=True).stdout)
