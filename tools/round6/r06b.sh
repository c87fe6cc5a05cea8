#!/bin/bash
# LDS bank conflicts of the Winograd kernels (rocprofv3 PMC, its own run) + stage A/B
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/r06b_pmc -o pmc --output-format csv -- python $R/tools/round6/wino_trace.py 16 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/r06b_pmc/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"][:90]
        if "wino" not in k and "split_kernel" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, d in agg.items():
    print(k, "launches", cnt[k], {c: round(v / max(cnt[k], 1)) for c, v in d.items()}, "conflict/active %.3f" % (d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1)))
PY
cd $R; bash tools/round6/r06a.sh
