#!/bin/bash
# raw single-pass PMC means per (kernel, grid): tools/pmc_raw.sh <tag> "<counter list>" [env assignments...]  (PMC_CMD, PMC_FILTER as pmc_quick.sh)
TAG=$1; LIST=$2; shift; shift
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
timeout 300 rocprofv3 --pmc $LIST --kernel-trace --output-format csv -d gpurun_out/q_$TAG -o q -- ${PMC_CMD:-python tools/vlad_trace32.py} > /dev/null 2> gpurun_out/q_$TAG.err
python - <<PY
import csv, glob, collections, os
FILTER = os.environ.get("PMC_FILTER", "vlad")
d = "gpurun_out/q_$TAG"
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[(r["Kernel_Name"][:58], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, g), m in cnt.items():
    if FILTER not in k: continue
    print(f"{k:58s} grid {g:8d} " + " ".join(f"{c.replace('SQ_', '')}={sum(v) / len(v):.4g}" for c, v in sorted(m.items())))
PY
rm -rf gpurun_out/q_$TAG
