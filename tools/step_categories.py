#!/usr/bin/env python3
"""What a step is made of, from a rocprofv3 (rocpd sqlite) kernel trace: the window between the last two launches of an anchor kernel (default
`k_solve_pose`: one per batch), its kernels summed by CATEGORY (ray-MLP chain, sweep, sample pre-pass, select bookkeeping, top-k, image side, copies,
other), the idle time of the window and the kernels that follow the longest gaps.  Written for the streamed scorer, whose step is thousands of launches
(VERDICT r4 #3c: "a quarter of the step is neither chain nor sweep").
Usage: step_categories.py results.db [out.md] [anchor-substring] [step-from-the-end] [start-anchor-substring]
(start-anchor: the window opens behind the LAST launch of that kernel before the end anchor instead of behind the previous end anchor -- for a run
with a single step, e.g. `k_emit_isocell` = the end of a streamed scene's set-up)"""
import re
import sqlite3
import sys

CATS = [
    ("ray-MLP chain (k_dense_planes, k_ray_encode_planes)", r"k_dense_planes|k_ray_encode_planes"),
    ("sweep (k_logits_f16x<.., 3, ..>)", r"k_logits_f16x<\d+, ?3"),
    ("sample pre-pass (k_logits_f16x<.., 2, ..>)", r"k_logits_f16x<\d+, ?2"),
    ("two-pass logits (k_logits_f16x<.., 0/1, ..>, k_score_reduce*)", r"k_logits_f16x<\d+, ?[01]|k_score_reduce"),
    ("select bookkeeping (k_split_q_slots, k_merge_stats*, k_sel_*, k_plane_norm_max, k_split_tiles_f16)", r"k_split_q_slots|k_merge_stats|k_sel_|k_plane_norm_max|k_split_tiles_f16"),
    ("top-k (k_topk_*)", r"k_topk_"),
    ("q_proj / dense helpers (k_linear*, k_q_proj, k_splitk*)", r"k_linear|k_q_proj|k_splitk"),
    ("pose solve (k_solve_pose)", r"k_solve_pose"),
    ("copies / fills (copyBuffer, fillBuffer, memset)", r"copyBuffer|fillBuffer|[Mm]emset|CatArrayBatchedCopy"),
    ("index / gather (index_elementwise, gather, arange ...)", r"index_elementwise|index_select|gather|arange|masked|where"),
]


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
    cs = "start" if "start" in cols else "start_timestamp"
    ce = "end" if "end" in cols else "end_timestamp"
    anchor = sys.argv[3] if len(sys.argv) > 3 else "k_solve_pose"
    back = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    rows = list(cur.execute(f'select name, "{cs}", "{ce}" from kernels order by "{cs}"'))
    marks = [i for i, r in enumerate(rows) if anchor in r[0]]
    start_anchor = sys.argv[5] if len(sys.argv) > 5 else None
    if start_anchor:
        if len(marks) < back:
            raise SystemExit(f"only {len(marks)} {anchor} launches in the trace")
        hi = marks[-back] + 1
        starts = [i for i, r in enumerate(rows[:hi]) if start_anchor in r[0]]
        if not starts:
            raise SystemExit(f"no {start_anchor} launch before the end anchor")
        lo, t0 = starts[-1] + 1, rows[starts[-1]][2]
    else:
        if len(marks) < back + 1:
            raise SystemExit(f"only {len(marks)} {anchor} launches in the trace")
        lo, hi = marks[-back - 1] + 1, marks[-back] + 1
        t0 = rows[marks[-back - 1]][2]
    step = rows[lo:hi]
    total = step[-1][2] - t0
    cat_t, cat_n = {}, {}
    other = {}
    prev_end, busy, gaps = t0, 0, {}
    for name, s, e in step:
        n = short(name)
        for label, pat in CATS:
            if re.search(pat, n):
                break
        else:
            label = "image side and everything else (PyTorch kernels: ViT, CNN, resize ...)"
            other[n] = other.get(n, 0) + (e - s)
        cat_t[label] = cat_t.get(label, 0) + (e - s)
        cat_n[label] = cat_n.get(label, 0) + 1
        gap = max(0, s - prev_end)
        gaps[n] = gaps.get(n, 0) + gap
        busy += e - max(s, prev_end) if e > prev_end else 0
        prev_end = max(prev_end, e)
    out = [f"step = {len(step)} kernel launches between two `{anchor}` launches: {total / 1e6:.2f} ms; GPU busy {busy / 1e6:.2f} ms, idle {(total - busy) / 1e6:.2f} ms "
           f"({100.0 * (total - busy) / total:.1f} %)", "", "| category | launches | kernel ms (summed) | % of the step |", "|---|---:|---:|---:|"]
    for label, t in sorted(cat_t.items(), key=lambda kv: -kv[1]):
        out.append(f"| {label} | {cat_n[label]} | {t / 1e6:.2f} | {100.0 * t / total:.1f} |")
    out.append(f"| idle (no kernel running) | | {(total - busy) / 1e6:.2f} | {100.0 * (total - busy) / total:.1f} |")
    out += ["", "idle time by the kernel that follows the gap (top 10):", ""]
    for n, g in sorted(gaps.items(), key=lambda kv: -kv[1])[:10]:
        out.append(f"* `{n}`: {g / 1e6:.2f} ms")
    if other:
        out += ["", "largest kernels of the last category:", ""]
        for n, t in sorted(other.items(), key=lambda kv: -kv[1])[:8]:
            out.append(f"* `{n}`: {t / 1e6:.2f} ms")
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 2 and sys.argv[2] != "-":
        open(sys.argv[2], "a").write(text)
    else:
        print(text)


if __name__ == "__main__":
    main()
