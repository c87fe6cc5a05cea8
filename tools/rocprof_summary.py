#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd SQLite output) kernel trace: per-kernel calls / total / avg / min / max.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r01 -- python bench.py ...     (on the GPU box)
    python tools/rocprof_summary.py gpurun_out/prof/r01_results.db > profiles/r01_kernel_stats.md
"""
import sqlite3
import sys


def main(path, label=""):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                      "max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(workgroup_x) "
                      "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats summary {label}\n")
    print(f"source: `{path}` -- total kernel time {tot / 1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches\n")
    print("| % | calls | total ms | avg us | min us | max us | vgpr | agpr | lds B | wg | kernel |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for n, c, t, a, mn, mx, vg, ag, lds, wg in rows:
        print(f"| {t / tot * 100:.1f} | {c} | {t / 1e6:.3f} | {a / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {vg} | {ag} | {lds} | {wg} | `{n[:120]}` |")


if __name__ == "__main__":
    main(sys.argv[1], " ".join(sys.argv[2:]))
