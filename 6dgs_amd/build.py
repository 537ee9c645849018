"""Build the HIP shared library (gfx950) in-tree: 6dgs_amd/csrc/lib6dgs_hip.so.

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "lib6dgs_hip.so")
HOSTCHECK = os.path.join(CSRC, "libsixdgs_hostcheck.so")
SOURCES = ["geometry.hip", "gemm.hip", "dense.hip", "score.hip", "pose.hip", "vit.hip"]
HEADERS = ["common.h", "device_math.h", "gemm_kernel.h", "dense.h", os.path.join("..", "..", "include", "sixdgs.h"), "dense_layout.h", "sweep_plan.h"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]
FLAGS += os.environ.get("SIXDGS_EXTRA_FLAGS", "").split()       # developer builds (e.g. -DSDG_DENSE_PROF: tools/prof_dense.py)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    if force or _stale(LIB, deps):
        objs = []
        procs = []
        for s in SOURCES:
            o = os.path.join(CSRC, s.replace(".hip", ".o"))
            objs.append(o)
            cmd = [HIPCC, *FLAGS, "-c", os.path.join(CSRC, s), "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        for cmd, p in procs:
            out, _ = p.communicate()
            if p.returncode != 0:
                sys.stderr.write(out)
                raise RuntimeError("hipcc failed: " + " ".join(cmd))
            if verbose and out.strip():
                print(out)
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        subprocess.check_call(cmd)
    return LIB


def build_hostcheck(force: bool = False) -> str:
    """Host instantiation of device_math.h and dense_layout.h for the CPU test-suite (no GPU code inside)."""
    src = os.path.join(CSRC, "hostcheck.cpp")
    deps = [src, os.path.join(CSRC, "device_math.h"), os.path.join(CSRC, "dense_layout.h"), os.path.join(CSRC, "sweep_plan.h")]
    if force or _stale(HOSTCHECK, deps):
        cmd = [HIPCC, "-x", "hip", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", src,
               "-o", HOSTCHECK]
        subprocess.check_call(cmd)
    return HOSTCHECK


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_hostcheck(force="--force" in sys.argv))
