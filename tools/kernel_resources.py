"""Print register / LDS / spill figures of the kernels of one HIP source (cross-compiles to gfx950 assembly)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "6dgs_amd", "csrc", "score.hip")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = os.path.join(ROOT, "gpurun_out", "asm")
os.makedirs(out, exist_ok=True)
asm = os.path.join(out, os.path.basename(src) + ".s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", *os.environ.get("SIXDGS_EXTRA_FLAGS", "").split(), "-S", "--cuda-device-only",
                       "-o", asm, src], stderr=subprocess.DEVNULL)
t = open(asm).read()
g = lambda blk, k: re.search(r"\." + k + r":\s+(\d+)", blk).group(1)
print(f"{'kernel':70s} agpr vgpr(total)  lds  sgpr_spill vgpr_spill")
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size", t, re.S):
    blk = m.group(0)
    nm = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if flt in nm:
        print(f"{nm[:70]:70s} {g(blk,'agpr_count'):>4s} {g(blk,'vgpr_count'):>6s} {g(blk,'group_segment_fixed_size'):>8s} {g(blk,'sgpr_spill_count'):>6s} {g(blk,'vgpr_spill_count'):>6s}")
