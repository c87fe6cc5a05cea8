#!/bin/bash
# round 4 evidence, part 2 (after profiles/r04y* are committed): full GPU suite + smoke + the default bench line + the driver-style line at HEAD,
# the 8-rank rehearsal on one GPU, and a kernel trace of a 200-key-frame region for the pipeline's GPU-busy fraction
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/r04z_pytest_gpu.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/r04z_pytest_gpu.log)"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r04z_smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $OUT/r04z_smoke.log)"
echo "t=$(( $(date +%s) - T0 ))s"
timeout 900 python bench.py > $OUT/r04z_bench.json 2> $OUT/r04z_bench.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/r04z_bench_driver_style.json 2> $OUT/r04z_bench_driver_style.err; echo "bench driver-style rc=$?"
echo "t=$(( $(date +%s) - T0 ))s"
LEGS="--no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --parity-steps 0 --c5-rows 0 --long-region-steps 0"
timeout 400 rocprofv3 --kernel-trace -d $OUT/r04z_trace -o r04z -- python bench.py --steps 200 --warmup 32 --min-time 0 $LEGS > $OUT/r04z_bench_under_rocprof.json 2> $OUT/r04z_trace.err
python tools/pipeline_busy.py $(ls $OUT/r04z_trace/*_results.db $OUT/r04z_trace/*/*_results.db 2>/dev/null | head -1) "r04z: HEAD, bench.py --steps 200 --warmup 32 (f16, 4 units in flight), under rocprofv3 --kernel-trace" > $OUT/r04z_pipeline_busy.json 2>> $OUT/r04z_trace.err
cat $OUT/r04z_pipeline_busy.json
find $OUT/r04z_trace -name '*.db' -size +20M -delete
bash tools/rehearse_ranks.sh 8 $OUT/r04z_bench_8ranks_one_gpu_stub.json --pipelines 2 > $OUT/r04z_rehearse.log 2>&1; echo "rehearse rc=$?"; tail -3 $OUT/r04z_rehearse.log | cut -c1-400
echo "t=$(( $(date +%s) - T0 ))s"
python - <<PY
import json
for f in ("$OUT/r04z_bench.json", "$OUT/r04z_bench_driver_style.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value", d["value"], "parity", (d.get("value_parity") or {}).get("value"), "geom", (d.get("with_geometry") or {}).get("value"), "f32", (d.get("value_f32") or {}).get("value"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "long_regions", (d.get("value_long_regions") or {}).get("value"))
        print(" db100k", {k: v.get("value") for k, v in (d.get("db100k") or {}).items()}, "c5", (d.get("c5_shard") or {}).get("value"), "python", (d.get("python_host") or {}).get("value"), "loop_match", d["loop_match"]["p50_ms"])
        print(" host", d.get("host_ms_per_microbatch"), "lat", d.get("keyframe_latency_ms", {}).get("p50"))
        print(" roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "frac_algorithmic", "launch_ms", "tiles_left_out", "traffic")}, d["roofline"]["rocprof_trace_note"][-12:], (d["roofline"]["rocprof_trace"] or {}).get("median_us"))
        rp = d.get("roofline_parity")
        if rp: print(" roofline_parity", {k: rp[k] for k in ("achieved", "frac", "frac_algorithmic", "launch_ms", "tiles_left_out", "traffic")}, rp["superpoint_ms_per_keyframe"], rp["rocprof_trace_note"][-12:], (rp["rocprof_trace"] or {}).get("median_us"))
    except Exception as e:
        print(f, "parse failed", e)
PY
