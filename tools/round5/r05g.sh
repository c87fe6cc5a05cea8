#!/bin/bash
# round 5, call 7: everything so far together -- the whole GPU suite, smoke, the N > 1 flow of bench.py rehearsed on one GPU (2 and 8 ranks through the
# stand-in for librccl: gloo plumbing group, per-rank keys, the all-gathers' device time), and the driver-style bench line
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/r05g_pytest_gpu.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/r05g_pytest_gpu.log)"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r05g_smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $OUT/r05g_smoke.log)"
echo "t=$(( $(date +%s) - T0 ))s"
for N in 2 8; do
  bash tools/rehearse_ranks.sh $N $OUT/r05g_bench_${N}ranks_one_gpu_stub.json --pipelines 2 > $OUT/r05g_rehearse$N.log 2>&1; echo "rehearse $N rc=$?"; tail -3 $OUT/r05g_rehearse$N.log | cut -c1-600
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$OUT/r05g_bench_${N}ranks_one_gpu_stub.json").read().strip().splitlines() if l.startswith("{")][-1])
    print({k: d.get(k) for k in ("n_gpus", "rccl_ranks", "torch_distributed_backend", "librccl", "librccl_mapped_in_process", "per_rank_keyframes_per_s", "db_rows_per_gpu", "all_gather_us_p50")})
except Exception as e:
    print("parse failed", e)
PY
done
echo "t=$(( $(date +%s) - T0 ))s"
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/r05g_bench_driver_style.json 2> $OUT/r05g_bench_driver_style.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("$OUT/r05g_bench_driver_style.json").read().strip().splitlines()[-1])
print("value", d["value"], "within", d["within_north_star_tolerance"], "value_within", d["value_within_north_star_tolerance"], "long", (d.get("value_long_regions") or {}).get("value"))
print("db100k", {k: (v.get("value") if isinstance(v, dict) else v) for k, v in (d.get("db100k") or {}).items() if k.endswith("rows")}, "c5", (d.get("c5_shard") or {}).get("value"), ((d.get("c5_shard") or {}).get("split") or {}).get("value"))
print("geom", (d.get("with_geometry") or {}).get("value"), "f32", (d.get("value_f32") or {}).get("value"), "py", (d.get("python_host") or {}).get("value"), "cpu", d["cpu_baseline"]["value"])
print("stages", d["roofline"]["stages_ms_per_keyframe"], d["roofline"]["frac"])
print("stages split", d["roofline_parity"]["stages_ms_per_keyframe"], d["roofline_parity"]["frac"])
PY
echo "t=$(( $(date +%s) - T0 ))s"
