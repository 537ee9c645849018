"""Round 6: which bit of a hipExtStreamCreateWithCUMask mask is which CU of which XCD on MI355X?  Launches one-per-CU workgroups (the sweep's LDS footprint,
spinning ~1 ms so that a whole grid is resident at once) on streams with different masks and prints, per XCD, how many DISTINCT CUs the workgroups ran on
and how many workgroups each XCD received.  python tools/probe_cumask.py"""
import collections
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def show(tag, xcc, cu, wgs_by_block=None):
    per = collections.defaultdict(set)
    cnt = collections.Counter()
    for x, c in zip(xcc.tolist(), cu.tolist()):
        per[x].add(c)
        cnt[x] += 1
    print(f"{tag}: " + "  ".join(f"xcd{x}: {cnt[x]} wgs on {len(per[x])} cus" for x in sorted(per)), flush=True)


def main():
    import torch
    ops = importlib.import_module("6dgs_amd.ops")
    torch.zeros(1, device="cuda")
    xcc, cu = ops.wg_placement(256)
    show("no mask, 256 wgs", xcc, cu)
    rr = (xcc == (np.arange(256) % 8)).mean()
    print(f"  workgroup i on XCD i % 8: {rr:.3f} of the workgroups", flush=True)
    full = [32] * 8
    cases = [("xcd-minor 31 x 8 (248)", ops.cu_mask_words([31] * 8, "xcd-minor"), 256),
             ("xcd-minor 28 on xcd 7 (252)", ops.cu_mask_words(full[:7] + [28], "xcd-minor"), 256),
             ("xcd-minor 24 on xcd 7 (248)", ops.cu_mask_words(full[:7] + [24], "xcd-minor"), 256),
             ("xcd-major 28 on xcd 7 (252)", ops.cu_mask_words(full[:7] + [28], "xcd-major"), 256),
             ("xcd-minor only xcd 3", ops.cu_mask_words([0, 0, 0, 32, 0, 0, 0, 0], "xcd-minor"), 64),
             ("low 32 bits only", [0xFFFFFFFF] + [0] * 7, 64)]
    for tag, words, n in cases:
        try:
            st = ops.cu_masked_stream("cuda:0", words)
            with torch.cuda.stream(st):
                xcc, cu = ops.wg_placement(n)
            torch.cuda.synchronize()
            show(tag + f", {n} wgs", xcc, cu)
        except Exception as e:      # noqa: BLE001
            print(tag, "failed:", e, flush=True)
    # timing: does a 256-workgroup grid on the 252-CU stream take two rounds (the 4 extra workgroups of XCD 7 wait for a CU)?
    for tag, words in (("no mask", None), ("28 on xcd 7", ops.cu_mask_words(full[:7] + [28], "xcd-minor"))):
        st = ops.cu_masked_stream("cuda:0", words) if words is not None else torch.cuda.Stream()
        with torch.cuda.stream(st):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ops.wg_placement(256)
            a.record(st)
            ops_out = torch.zeros(256, dtype=torch.int32, device="cuda")
            from importlib import import_module
            lib = import_module("6dgs_amd._lib").load()
            import ctypes as C
            lib.sixdgs_debug_wg_placement(256, 2_000_000, C.c_void_p(ops_out.data_ptr()), C.c_void_p(st.cuda_stream))
            b.record(st)
        torch.cuda.synchronize()
        print(f"256 workgroups of 2 M cycles on '{tag}': {a.elapsed_time(b):.3f} ms", flush=True)


if __name__ == "__main__":
    main()
