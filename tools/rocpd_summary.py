#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace: per-kernel calls / total / average /
share, like `--stats`, plus launch geometry of the top kernels.  Usage: rocpd_summary.py results.db [out.md]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:<>, ]+?)\(", name)
    name = m.group(1) if m else name
    return name if len(name) <= 90 else name[:87] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = ["| kernel | calls | total ms | avg ms | % of GPU kernel time |", "|---|---:|---:|---:|---:|"]
    for name, calls, tot, avg, pct in rows[:40]:
        lines.append(f"| `{short(name)}` | {calls} | {tot / 1e3:.3f} | {avg / 1e3:.4f} | {pct:.2f} |")   # rocpd durations are in us
    geo = ["", "| kernel | grid | workgroup | VGPR | AGPR | SGPR | static LDS B | scratch B |", "|---|---|---|---:|---:|---:|---:|---:|"]
    seen = set()
    for r in cur.execute("select name, grid_x, grid_y, grid_z, workgroup_x, workgroup_y, workgroup_z, vgpr_count, accum_vgpr_count, "
                         "sgpr_count, static_lds_size, scratch_size from kernels order by duration desc"):
        n = short(r[0])
        if n in seen or not ("k_" in n):
            continue
        seen.add(n)
        geo.append(f"| `{n}` | {r[1]}x{r[2]}x{r[3]} | {r[4]}x{r[5]}x{r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]} |")
        if len(seen) >= 14:
            break
    out = "\n".join(lines + geo) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "a").write(out)
    else:
        print(out)


if __name__ == "__main__":
    main()
