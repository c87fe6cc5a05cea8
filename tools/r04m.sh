#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TAG=${1:-r04m}
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/${TAG}_pytest_gpu.log)"
timeout 600 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("$OUT/${TAG}_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "parity", (d.get("value_parity") or {}).get("value"), "geom", (d.get("with_geometry") or {}).get("value"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
print(" roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "frac_algorithmic", "launch_ms", "tiles_left_out")})
rp = d.get("roofline_parity")
if rp: print(" roofline_parity", {k: rp[k] for k in ("achieved", "frac", "frac_algorithmic", "launch_ms", "tiles_left_out")}, rp["stages_ms_per_keyframe"])
PY
