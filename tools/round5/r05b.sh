#!/bin/bash
# round 5, call 2: the host-side changes on hardware -- full-size index tests against the blocked oracle, the streaming intake's latency bound,
# the shard tests (+ exchange timing), the e2e tests through push_keyframe, and a quick bench line with the new keys
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_index.py tests/test_gpu_shard_rccl.py tests/test_gpu_e2e_depth.py tests/test_gpu_e2e_scene.py -m gpu -q -x > $OUT/r05b_pytest_a.log 2>&1; echo "pytest a rc=$? $(tail -1 $OUT/r05b_pytest_a.log)"
timeout 900 python -m pytest tests/test_gpu_bench_shape.py -m gpu -q -x -k "host_loop or streaming or row_stride" > $OUT/r05b_pytest_b.log 2>&1; echo "pytest b rc=$? $(tail -1 $OUT/r05b_pytest_b.log)"
echo "t=$(( $(date +%s) - T0 ))s"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r05b_smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $OUT/r05b_smoke.log)"
timeout 600 python bench.py --steps 20 --warmup 5 --python-steps 0 --f32-steps 0 --geometry-steps 0 > $OUT/r05b_bench_driver_style.json 2> $OUT/r05b_bench_driver_style.err; echo "bench rc=$?"
echo "t=$(( $(date +%s) - T0 ))s"
python - <<PY
import json
d = json.loads(open("$OUT/r05b_bench_driver_style.json").read().strip().splitlines()[-1])
print("value", d["value"], "within", d["within_north_star_tolerance"], "value_within", d["value_within_north_star_tolerance"], "long", (d.get("value_long_regions") or {}).get("value"))
print("gate", d["parity_gate"])
print("db100k", {k: (v.get("value") if isinstance(v, dict) else v) for k, v in (d.get("db100k") or {}).items()})
print("c5", (d.get("c5_shard") or {}).get("value"), ((d.get("c5_shard") or {}).get("split") or {}).get("value"))
print("units", d["config"]["pipelines_per_gpu"], d["config"]["units_oldest_first"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"][:40])
PY
