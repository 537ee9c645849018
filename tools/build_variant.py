"""Private builds of lib6dgs_hip.so with extra compiler flags, for A/B timing on the GPU box (loaded through SIXDGS_LIB):
    python tools/build_variant.py <name> [flags...]   ->  build/variants/lib_<name>.so
build/ is git-ignored but travels with a gpurun snapshot."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, flags = sys.argv[1], sys.argv[2:]
csrc = os.path.join(ROOT, "6dgs_amd", "csrc")
out = os.path.join(ROOT, "build", "variants", f"lib_{name}.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function", *flags]
objs, procs = [], []
for f in ("geometry.hip", "gemm.hip", "dense.hip", "score.hip", "pose.hip", "vit.hip"):
    o = os.path.join(ROOT, "build", "variants", f"{name}_{f[:-4]}.o")
    objs.append(o)
    procs.append(subprocess.Popen(base + ["-c", os.path.join(csrc, f), "-o", o]))
for p in procs:
    if p.wait() != 0:
        raise SystemExit("hipcc failed")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out])
for o in objs:
    os.remove(o)
print(out)
