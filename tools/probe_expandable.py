"""Does PyTorch-ROCm's caching allocator support expandable segments here, and what do they do to the scene-to-scene pattern of the evaluation sweep
(a 100-200 GB plane buffer freed, small tensors allocated, the next scene's larger buffer asked for)?  Run with and without
PYTORCH_HIP_ALLOC_CONF=expandable_segments:True / PYTORCH_CUDA_ALLOC_CONF=..."""
import os, sys, time
import torch
dev = "cuda"
def gib(x): return round(x / 2**30, 1)
def state(tag):
    torch.cuda.synchronize()
    print(f"{tag}: allocated {gib(torch.cuda.memory_allocated())} reserved {gib(torch.cuda.memory_reserved())} free {gib(torch.cuda.mem_get_info()[0])} GiB", flush=True)
print("conf:", os.environ.get("PYTORCH_HIP_ALLOC_CONF"), os.environ.get("PYTORCH_CUDA_ALLOC_CONF"))
sizes = [100, 60, 140, 30, 170, 90]
small = []
for i, g in enumerate(sizes):
    t0 = time.perf_counter()
    big = torch.empty(g << 30, dtype=torch.uint8, device=dev)
    big[:: 1 << 21].zero_()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    small += [torch.empty(3 << 20, dtype=torch.uint8, device=dev) for _ in range(20)]       # survivors of the scene: small tensors inside the freed block
    ws = torch.empty(int(0.1 * g) << 30, dtype=torch.uint8, device=dev)
    state(f"scene {i}: {g} GiB planes in {t1 - t0:.2f} s")
    del big, ws
    if "--empty-cache" in sys.argv:
        torch.cuda.empty_cache()
state("end")
