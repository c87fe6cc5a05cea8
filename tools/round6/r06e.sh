#!/bin/bash
# the driver-style line (what BENCH records) with OMNI_SPLIT_WINO at its default, and a check of the wino masks
mkdir -p gpurun_out
python tools/round6/wino_check.py 2>&1 | grep -E "mask (7)|FAILED"
python bench.py --steps 20 --warmup 5 > gpurun_out/r06e_bench_driver_style.json 2> gpurun_out/r06e_err.log
python - <<PY
import json
d=json.loads(open("gpurun_out/r06e_bench_driver_style.json").read().strip().split("\n")[-1])
print("value", d["value"], "value_parity", d.get("value_parity"), "long", (d.get("value_long_regions") or {}).get("value"))
print("cpu_baseline", {k: d["cpu_baseline"][k] for k in ("value","cores","ms_per_keyframe","superpoint_ms_per_image","sweep","cpu_model") if k in d["cpu_baseline"]})
print("parity_split", d.get("parity_split"))
print("roofline_parity", d.get("roofline_parity"))
PY
