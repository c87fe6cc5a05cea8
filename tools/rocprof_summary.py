#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd SQLite output) kernel trace: per-kernel calls / total / avg / median / min / max.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r01 -- python bench.py ...     (on the GPU box)
    python tools/rocprof_summary.py gpurun_out/prof/r01_results.db [--json profiles/r01_kernel_times.json] [label ...] > profiles/r01_kernel_stats.md

The median matters for the kernels bench.py quotes a roofline for: a handle's calibration pass (one image) launches the same kernel as the
64-image launches of the timed loop, so the average mixes two launch shapes; the median is the timed loop's launch.  The JSON side file
(kernel -> calls / avg / median us) is what bench.py's `rocprof_trace` field cites."""
import json
import sqlite3
import sys
from collections import defaultdict


def main(path, label="", json_out=None):
    db = sqlite3.connect(path)
    per = defaultdict(list)
    meta = {}
    for n, d, vg, ag, lds, wg in db.execute("select name, end-start, vgpr_count, accum_vgpr_count, lds_size, workgroup_x from kernels"):
        per[n].append(d)
        m = meta.setdefault(n, [0, 0, 0, 0])
        m[0], m[1], m[2], m[3] = max(m[0], vg or 0), max(m[1], ag or 0), max(m[2], lds or 0), max(m[3], wg or 0)
    rows = sorted(((n, sorted(v)) for n, v in per.items()), key=lambda r: -sum(r[1]))
    tot = sum(sum(v) for _, v in rows)
    print(f"# rocprofv3 --kernel-trace --stats summary {label}\n")
    print(f"source: `{path}` -- total kernel time {tot / 1e6:.2f} ms over {sum(len(v) for _, v in rows)} dispatches\n")
    print("| % | calls | total ms | avg us | median us | min us | max us | vgpr | agpr | lds B | wg | kernel |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    out = {}
    for n, v in rows:
        t, c = sum(v), len(v)
        med = v[c // 2] if c % 2 else 0.5 * (v[c // 2 - 1] + v[c // 2])
        vg, ag, lds, wg = meta[n]
        print(f"| {t / tot * 100:.1f} | {c} | {t / 1e6:.3f} | {t / c / 1e3:.2f} | {med / 1e3:.2f} | {v[0] / 1e3:.2f} | {v[-1] / 1e3:.2f} | {vg} | {ag} | {lds} | {wg} | `{n[:120]}` |")
        out[n] = {"calls": c, "avg_us": round(t / c / 1e3, 2), "median_us": round(med / 1e3, 2), "min_us": round(v[0] / 1e3, 2), "max_us": round(v[-1] / 1e3, 2)}
    if json_out:
        json.dump({"label": label, "source": path, "kernels": out}, open(json_out, "w"), indent=1)


if __name__ == "__main__":
    a = sys.argv[1:]
    jo = None
    if "--json" in a:
        i = a.index("--json")
        jo = a[i + 1]
        del a[i:i + 2]
    main(a[0], " ".join(a[1:]), jo)
