#!/bin/bash
# checkpoint at HEAD: the whole GPU suite, smoke(), the driver-style line
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r06z_pytest_gpu_full.log 2>&1; grep -E "passed|failed|error" gpurun_out/r06z_pytest_gpu_full.log | tail -5 | tee gpurun_out/r06z_pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke() returned')" > gpurun_out/r06z_smoke_full.log 2>&1; echo "smoke rc=$?" | tee gpurun_out/r06z_smoke.log; grep -iE "smoke|parity|ok" gpurun_out/r06z_smoke_full.log | tail -5 | tee -a gpurun_out/r06z_smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06z_bench_driver_style.json 2> gpurun_out/r06z_err.log
python - <<PY
import json
d=json.loads(open("gpurun_out/r06z_bench_driver_style.json").read().strip().split("\n")[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "value_parity", (d.get("value_parity") or {}).get("value"), "long", (d.get("value_long_regions") or {}).get("value"))
print("roofline", {k: d["roofline"].get(k) for k in ("achieved","frac","launch_ms")}, d["roofline"]["mask_skip"].get("mobilenetvlad_tiles_left_out"))
print("stages", d["roofline"]["stages_ms_per_keyframe"])
print("parity_gate", d.get("parity_gate"))
rp = d.get("roofline_parity") or {}
print("roofline_parity", {k: rp.get(k) for k in ("achieved","frac","launch_ms","frac_algorithmic")})
PY
tail -3 gpurun_out/r06z_err.log
