#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv): per kernel, per counter, mean value per dispatch.

    python tools/pmc_summary.py gpurun_out/r01c_pmc1 gpurun_out/r01c_pmc2 ... > profiles/r01c_pmc_summary.md

Each directory is one --pmc pass (counters cannot all share a pass: SQ 8 slots, TCC 4, FETCH_SIZE costs 3, WRITE_SIZE 2).
Units (calibrated on conv3x3_c64: SQ_VALU_MFMA_BUSY_CYCLES = 32 x #v_mfma_f32_32x32x16_f16 over the whole chip, while
GRBM_GUI_ACTIVE is summed over the 8 XCDs): MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)
(GRBM_GUI_ACTIVE comes from a different pass than the SQ counters: same workload, same dispatch mix).
gfx950 note (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read; the
`hbm_read_MB (x2)` column applies that correction; FETCH_SIZE/WRITE_SIZE are in KiB.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("omni::", "")
    cut = name.find("(")
    return (name[:cut] if cut > 0 else name)[:70]


def main(dirs):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))     # kernel -> counter -> [sum, n]
    dur = defaultdict(lambda: [0.0, 0])
    dirs = [d for d in dirs if os.path.isdir(d)]
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = short(row.get("Kernel_Name", "?"))
                    c = row.get("Counter_Name", "?")
                    v = float(row.get("Counter_Value", 0) or 0)
                    a = acc[k][c]
                    a[0] += v
                    a[1] += 1
                    if "Start_Timestamp" in row and row["Start_Timestamp"]:
                        dd = dur[(k, c)]
                        dd[0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                        dd[1] += 1
    counters = sorted({c for k in acc for c in acc[k]})
    print("# rocprofv3 --pmc summary: mean counter value per dispatch\n")
    print("sources: " + ", ".join(f"`{d}`" for d in dirs) + "\n")
    print("| kernel | dispatches | " + " | ".join(counters) + " | hbm_read_MB (FETCH_SIZE x2) | hbm_write_MB | MFMA util % (= MFMA_BUSY / (GUI_ACTIVE x 128)) |")
    print("|---|---|" + "---|" * (len(counters) + 3))
    for k in sorted(acc, key=lambda k: -max(a[1] for a in acc[k].values())):
        n = max(a[1] for a in acc[k].values())
        mean = {c: (acc[k][c][0] / acc[k][c][1] if acc[k][c][1] else None) for c in counters}
        cells = ["" if mean[c] is None else f"{mean[c]:.4g}" for c in counters]
        rd = f"{mean['FETCH_SIZE'] * 1024 * 2 / 1e6:.2f}" if mean.get("FETCH_SIZE") is not None else ""
        wr = f"{mean['WRITE_SIZE'] * 1024 / 1e6:.2f}" if mean.get("WRITE_SIZE") is not None else ""
        mf = ""
        if mean.get("SQ_VALU_MFMA_BUSY_CYCLES") and mean.get("GRBM_GUI_ACTIVE"):
            mf = f"{100.0 * mean['SQ_VALU_MFMA_BUSY_CYCLES'] / (mean['GRBM_GUI_ACTIVE'] * 128.0):.1f}"
        print(f"| `{k}` | {n} | " + " | ".join(cells) + f" | {rd} | {wr} | {mf} |")


if __name__ == "__main__":
    main(sys.argv[1:])
