#!/bin/bash
# round 4, last call: the end-to-end tests with the two-half upload, then the default and the driver-style bench lines at the last commit
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_e2e_scene.py tests/test_gpu_e2e_depth.py tests/test_gpu_shard_rccl.py -x -q -m gpu > gpurun_out/r04ag_pytest.log 2>&1
echo "pytest rc=$? $(grep -E 'passed|failed' gpurun_out/r04ag_pytest.log | tail -1)"
timeout 900 python bench.py > gpurun_out/r04ag_bench.json 2> gpurun_out/r04ag_bench.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04ag_bench_driver_style.json 2> gpurun_out/r04ag_bench_driver_style.err; echo "bench driver-style rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/r04ag_bench.json", "gpurun_out/r04ag_bench_driver_style.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "value", d["value"], "parity", d["value_parity"]["value"], "geom", d["with_geometry"]["value"], "long", (d.get("value_long_regions") or {}).get("value"), "roofline", d["roofline"]["frac"], d["roofline_parity"]["frac"], "lat", d["keyframe_latency_ms"]["p50"])
PY
