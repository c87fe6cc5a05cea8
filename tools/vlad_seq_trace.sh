#!/bin/bash
# per-dispatch kernel durations of ONE MobileNetVLAD forward at BATCH images, in launch order: tools/vlad_seq_trace.sh <f16|f32> [batch]
export TMPDIR=/tmp PREC=${1:-f16} BATCH=${2:-32} ORACLE=0
rm -rf gpurun_out/vs
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/vs -o vs -- python tools/vlad_trace32.py > /dev/null 2>&1
python - <<'PY'
import sqlite3
db = sqlite3.connect("gpurun_out/vs/vs_results.db")
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else "0")
print("columns:", cols)
rows = db.execute(f"select name, end-start, {gx}, workgroup_x, lds_size, vgpr_count from kernels order by start").fetchall()
# one forward = from a stem kernel to the next one
idx = [i for i, r in enumerate(rows) if "stem" in r[0]]
seq = rows[idx[-2]:idx[-1]]
tot = 0
for n, d, g, wg, lds, vg in seq:
    tot += d
    print(f"{d / 1e3:8.2f} us  wgs {g // max(wg, 1):6d}  lds {lds:6d}  vgpr {vg:3d}  {n[:90]}")
print(f"sum {tot / 1e3:.1f} us over {len(seq)} dispatches")
PY
rm -rf gpurun_out/vs
