#!/bin/bash
# quick single-pass PMC probe: tools/pmc_quick.sh <tag> [env assignments...]   (PMC_CMD = command to profile, PMC_FILTER = kernel-name substring;
# defaults: a short bench.py run, "conv")
TAG=$1; shift
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --kernel-trace --output-format csv -d gpurun_out/q_$TAG -o q -- ${PMC_CMD:-python bench.py --steps 4 --warmup 1 --no-cpu-baseline --match-db-rows 4096 --big-db-keyframes 0 --f32-steps 0 --python-steps 0 --geometry-steps 0 --batched-rows 0} > /dev/null 2> gpurun_out/q_$TAG.err
python - <<PY
import csv, glob, collections, os
FILTER = os.environ.get("PMC_FILTER", "conv")
d = "gpurun_out/q_$TAG"
dur = collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Kernel_Name"]][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["Grid_Size"])))
print("tag $TAG")
for k in cnt:
    if FILTER not in k: continue
    # split by grid size (layers differ)
    grids = sorted({g for v in cnt[k].values() for _, g in v})
    for g in grids:
        m = {c: sum(x for x, gg in v if gg == g) / max(1, sum(1 for x, gg in v if gg == g)) for c, v in cnt[k].items()}
        cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
        line = f"{k[:60]:60s} grid {g:7d} cycles {cyc:9.0f} mfma_util {m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(cyc*1024+1e-9)*100:5.1f}%"
        wc = m.get("SQ_WAVE_CYCLES", 1)
        line += f" wave_cyc {wc:.3g} active {m.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} wait_inst {m.get('SQ_WAIT_INST_ANY',0)/wc:.2f} wait_any {m.get('SQ_WAIT_ANY',0)/wc:.2f} wait_lds {m.get('SQ_WAIT_INST_LDS',0)/wc:.2f} lds_conf/idx {m.get('SQ_LDS_BANK_CONFLICT',0)/(m.get('SQ_LDS_IDX_ACTIVE',1)+1e-9):.3f}"
        print(line)
    ds = dur.get(k, [])
    print("   durations us:", sorted(round(x / 1e3, 1) for x in ds)[:40])
PY
rm -rf gpurun_out/q_$TAG
