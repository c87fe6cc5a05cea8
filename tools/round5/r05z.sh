#!/bin/bash
# round 5 evidence, part 2 (after profiles/r05y* are committed: bench.py cites them): the whole GPU suite + smoke, the default bench line and the
# driver-style line at HEAD, a kernel trace of a 200-key-frame region for the pipeline's GPU-busy fraction, the 8-rank rehearsal on one GPU
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/r05z_pytest_gpu.log 2>&1; echo "pytest rc=$? $(grep -E 'passed|failed' $OUT/r05z_pytest_gpu.log | tail -1)"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r05z_smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $OUT/r05z_smoke.log)"
echo "t=$(( $(date +%s) - T0 ))s"
timeout 1200 python bench.py > $OUT/r05z_bench.json 2> $OUT/r05z_bench.err; echo "bench rc=$?"
echo "t=$(( $(date +%s) - T0 ))s"
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/r05z_bench_driver_style.json 2> $OUT/r05z_bench_driver_style.err; echo "bench driver-style rc=$?"
echo "t=$(( $(date +%s) - T0 ))s"
LEGS="--no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --parity-steps 0 --c5-rows 0 --long-region-steps 0"
timeout 400 rocprofv3 --kernel-trace -d $OUT/r05z_trace -o r05z -- python bench.py --steps 200 --warmup 32 --min-time 0 $LEGS > $OUT/r05z_bench_under_rocprof.json 2> $OUT/r05z_trace.err
python tools/pipeline_busy.py $(ls $OUT/r05z_trace/*_results.db $OUT/r05z_trace/*/*_results.db 2>/dev/null | head -1) "r05z: HEAD, bench.py --steps 200 --warmup 32 (f16, 4 units in flight), under rocprofv3 --kernel-trace" > $OUT/r05z_pipeline_busy.json 2>> $OUT/r05z_trace.err
cat $OUT/r05z_pipeline_busy.json
find $OUT/r05z_trace -name '*.db' -size +20M -delete
bash tools/rehearse_ranks.sh 8 $OUT/r05z_bench_8ranks_one_gpu_stub.json --pipelines 2 > $OUT/r05z_rehearse.log 2>&1; echo "rehearse rc=$?"; tail -2 $OUT/r05z_rehearse.log | cut -c1-300
echo "t=$(( $(date +%s) - T0 ))s"
python - <<PY
import json
for f in ("$OUT/r05z_bench.json", "$OUT/r05z_bench_driver_style.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value", d["value"], "within", d["within_north_star_tolerance"], "value_within", d["value_within_north_star_tolerance"], "parity", (d.get("value_parity") or {}).get("value"), "geom", (d.get("with_geometry") or {}).get("value"), "f32", (d.get("value_f32") or {}).get("value"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "long_regions", (d.get("value_long_regions") or {}).get("value"))
        print(" db100k", {k: v.get("value") for k, v in (d.get("db100k") or {}).items() if isinstance(v, dict)}, "c5", (d.get("c5_shard") or {}).get("value"), ((d.get("c5_shard") or {}).get("split") or {}).get("value"), "python", (d.get("python_host") or {}).get("value"), "loop_match", d["loop_match"]["p50_ms"])
        print(" host", d.get("host_ms_per_microbatch"), "lat", d.get("keyframe_latency_ms", {}).get("p50"), "units", d["config"]["pipelines_per_gpu"], d["config"]["units_oldest_first"])
        print(" roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "frac_algorithmic", "launch_ms", "tiles_left_out", "traffic")}, d["roofline"]["rocprof_trace_note"][-12:], (d["roofline"]["rocprof_trace"] or {}).get("median_us"), (d["roofline"]["rocprof_trace"] or {}).get("source"))
        print(" stages", d["roofline"]["stages_ms_per_keyframe"], d["roofline"]["superpoint_ms_per_keyframe"])
        rp = d.get("roofline_parity")
        if rp: print(" roofline_parity", {k: rp[k] for k in ("achieved", "frac", "frac_algorithmic", "launch_ms", "tiles_left_out", "traffic")}, rp["superpoint_ms_per_keyframe"], rp["rocprof_trace_note"][-12:], (rp["rocprof_trace"] or {}).get("median_us")); print(" stages split", rp["stages_ms_per_keyframe"])
        print(" knn", d["roofline_knn"]["frac"], (d.get("roofline_knn_batched") or {}).get("frac"), "gate", d["parity_gate"])
    except Exception as e:
        print(f, "parse failed", e)
PY
