#!/usr/bin/env python
"""Per timed region of a short-region bench run under rocprofv3 --kernel-trace: the kernel span, its busy union, the idle gaps inside it and the idle time
between regions -- where a 20-key-frame region (2.5 units) loses against the steady state.   python tools/region_timeline.py <results.db> [gap_us=400]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
gap_ns = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 400e3
rows = db.execute("select start, end, name from kernels order by start").fetchall()
regions, cur = [], []
last_end = None
for s, e, n in rows:
    if last_end is not None and s - last_end > gap_ns and cur:
        regions.append(cur); cur = []
    cur.append((s, e, n))
    last_end = e if last_end is None else max(last_end, e)
if cur: regions.append(cur)
out = []
for r in regions:
    if len(r) < 100: continue
    s0, e1 = r[0][0], max(x[1] for x in r)
    busy, ce = 0, s0
    gaps = []
    for s, e, n in r:
        if s > ce:
            gaps.append((s - ce, n))
            ce = s
        if e > ce:
            busy += e - ce; ce = e
    stems = sum(1 for x in r if "stem" in x[2])
    out.append((e1 - s0, busy, len(r), stems, sorted(gaps, reverse=True)[:4], r[0][2][:40], r[-1][2][:40]))
import statistics
full = [o for o in out if o[3] == statistics.mode([x[3] for x in out])]
print(f"{len(out)} regions, {len(full)} with the modal number of units ({full[0][3] if full else 0} stem launches)")
for o in full[len(full) // 2: len(full) // 2 + 3]:
    print(f"span {o[0] / 1e6:.3f} ms  busy {o[1] / 1e6:.3f} ms ({o[1] / o[0] * 100:.1f} %)  kernels {o[2]}  first {o[5]}  last {o[6]}")
    print("   largest idle gaps (us, next kernel):", [(round(g / 1e3, 1), n[:30]) for g, n in o[4]])
print(f"median span {statistics.median(o[0] for o in full) / 1e6:.3f} ms, median busy {statistics.median(o[1] for o in full) / 1e6:.3f} ms")
