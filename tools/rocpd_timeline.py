#!/usr/bin/env python3
"""Timeline of the LAST step of a traced bench.py run from a rocprofv3 (rocpd sqlite) kernel trace: every kernel launch between the last two
`k_solve_pose` launches with its start offset, duration and the idle gap before it; then the busy / idle split of the step and the gaps by the
kernel that follows them.  Usage: rocpd_timeline.py results.db [out.md] [anchor-kernel-substring] [step]
(step: which pair of anchors, counted from the end: 1 = the last step (default), 8 = the eighth from last -- a bench.py --graph run ends with five EAGER
steps for its HIP-event figures, so its replays are further back).  With a --hip-trace database it also lists the HIP API calls that overlap the step."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:<>, ]+?)\(", name)
    name = m.group(1) if m else name
    return name if len(name) <= 70 else name[:67] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    c_start = "start" if "start" in cols else ("start_timestamp" if "start_timestamp" in cols else None)
    c_end = "end" if "end" in cols else ("end_timestamp" if "end_timestamp" in cols else None)
    if c_start is None or c_end is None:
        raise SystemExit(f"no start/end columns in the kernels view: {cols}")
    anchor = sys.argv[3] if len(sys.argv) > 3 else "k_solve_pose"
    rows = list(cur.execute(f'select name, "{c_start}", "{c_end}" from kernels order by "{c_start}"'))
    marks = [i for i, r in enumerate(rows) if anchor in r[0]]
    if len(marks) < 2:
        raise SystemExit(f"fewer than two {anchor} launches in the trace")
    back = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    if len(marks) < back + 1:
        raise SystemExit(f"only {len(marks)} {anchor} launches in the trace")
    lo, hi = marks[-back - 1] + 1, marks[-back] + 1
    step = rows[lo:hi]
    t0 = rows[marks[-back - 1]][2]
    out = [f"last step: {len(step)} kernel launches between two `{anchor}` launches, {(step[-1][2] - t0) / 1e6:.3f} ms from the end of the previous step's {anchor} to the end of this one's", "",
           "| # | kernel | start ms | dur us | gap before us |", "|---:|---|---:|---:|---:|"]
    prev_end, busy, gaps = t0, 0, {}
    for i, (name, s, e) in enumerate(step):
        gap = max(0, s - prev_end)
        busy += e - max(s, prev_end) if e > prev_end else 0
        n = short(name)
        gaps[n] = gaps.get(n, 0) + gap
        if i < 400:
            out.append(f"| {i} | `{n}` | {(s - t0) / 1e6:.3f} | {(e - s) / 1e3:.1f} | {gap / 1e3:.1f} |")
        prev_end = max(prev_end, e)
    total = prev_end - t0
    out += ["", f"busy {busy / 1e6:.3f} ms, idle {(total - busy) / 1e6:.3f} ms of {total / 1e6:.3f} ms", "", "idle time by the kernel that follows the gap (top 12):", ""]
    for n, g in sorted(gaps.items(), key=lambda kv: -kv[1])[:12]:
        out.append(f"* `{n}`: {g / 1e3:.1f} us")
    # host side (only in a --hip-trace database): API calls that overlap the step's window, by total time
    try:
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
        view = next((v for v in ("regions", "region", "api_calls", "hip_api") if v in tabs), None)
        if view:
            rc = [r[1] for r in cur.execute(f"pragma table_info({view})")]
            cs = "start" if "start" in rc else ("start_timestamp" if "start_timestamp" in rc else None)
            ce = "end" if "end" in rc else ("end_timestamp" if "end_timestamp" in rc else None)
            if cs and ce and "name" in rc:
                w0, w1 = t0 - 30_000_000, prev_end
                api = {}
                for name, s0, e0 in cur.execute(f'select name, "{cs}", "{ce}" from {view} where "{ce}" >= ? and "{cs}" <= ?', (w0, w1)):
                    a = api.setdefault(name, [0, 0, 0])
                    a[0] += 1; a[1] += e0 - s0; a[2] = max(a[2], e0 - s0)
                out += ["", f"HIP API calls overlapping [step start - 30 ms, step end] ({view} view; calls, total ms, longest ms):", ""]
                for name, a in sorted(api.items(), key=lambda kv: -kv[1][1])[:14]:
                    out.append(f"* `{name}`: {a[0]} calls, {a[1] / 1e6:.3f} ms, longest {a[2] / 1e6:.3f} ms")
        else:
            out += ["", f"(no API view in this database; tables: {', '.join(tabs[:30])})"]
    except Exception as e:
        out += ["", f"(API listing failed: {e!r})"]
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 2 and sys.argv[2] != "-":
        open(sys.argv[2], "a").write(text)
    else:
        print(text)


if __name__ == "__main__":
    main()
