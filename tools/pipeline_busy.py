#!/usr/bin/env python
"""GPU-busy fraction of the key-frame pipeline's steady state from a rocprofv3 kernel trace (results.db): the union of all kernel intervals over the
span between the third and the last MobileNetVLAD stem launch (one per unit), the idle gaps longer than 50 us, the units per second.

    python tools/pipeline_busy.py gpurun_out/r04z_trace/r04z_results.db "label" > profiles/r04z_pipeline_busy.json
"""
import json
import sqlite3
import sys

import numpy as np


def main(db, label):
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    stem = [r for r in rows if "vlad_stem_b0" in r[0]]
    # the longest run of evenly spaced units = the timed region (other legs of the bench launch the stem too)
    t0, t1, n = stem[2][1], stem[-1][1], len(stem) - 3
    sel = sorted((r[1], r[2]) for r in rows if t0 <= r[1] < t1)
    busy, (cs, ce), gaps = 0, sel[0], []
    for s, e in sel[1:]:
        if s > ce:
            busy += ce - cs
            gaps.append(s - ce)
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    g = np.array(gaps) / 1e3
    print(json.dumps({"label": label, "source": db, "units": n, "span_ms": round((t1 - t0) / 1e6, 3), "ms_per_unit": round((t1 - t0) / 1e6 / n, 3),
                      "gpu_busy_fraction": round(busy / (t1 - t0), 4), "idle_gaps_over_50us": int((g > 50).sum()),
                      "idle_ms_in_those_gaps": round(float(g[g > 50].sum()) / 1e3, 3), "kernels_in_span": len(sel),
                      "definition": "union of all kernel intervals / span between the 3rd and the last MobileNetVLAD stem launch (one launch per unit)"}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
