#!/bin/bash
# round 5, last call: the whole GPU suite + smoke + the driver-style bench line at the final commit
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/r05zz_pytest_gpu.log 2>&1; echo "pytest rc=$? $(grep -E 'passed|failed' $OUT/r05zz_pytest_gpu.log | tail -1)"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r05zz_smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $OUT/r05zz_smoke.log)"
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/r05zz_bench_driver_style.json 2> $OUT/r05zz_bench_driver_style.err; echo "bench driver-style rc=$?"
python - <<PY
import json
d = json.loads(open("$OUT/r05zz_bench_driver_style.json").read().strip().splitlines()[-1])
print("value", d["value"], "within", d["within_north_star_tolerance"], "value_within", d["value_within_north_star_tolerance"], "long", (d.get("value_long_regions") or {}).get("value"), "lat", d["keyframe_latency_ms"])
PY
