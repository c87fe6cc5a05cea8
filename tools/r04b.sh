#!/bin/bash
# round 4, GPU call b: full GPU suite + smoke + the default bench line at HEAD (split mask skip on by default) + kernel trace / PMC passes (f16 and split)
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/${1}_pytest_gpu.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/${1}_pytest_gpu.log)"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/${1}_smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $OUT/${1}_smoke.log)"
timeout 600 python bench.py > $OUT/${1}_bench.json 2> $OUT/${1}_bench.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/${1}_bench_driver_style.json 2> $OUT/${1}_bench_driver_style.err; echo "bench driver-style rc=$?"
echo "t=$(( $(date +%s) - T0 ))s"
bash tools/profile_gpu.sh ${1}f
echo "t=$(( $(date +%s) - T0 ))s"
bash tools/profile_gpu.sh ${1}s --precision split
echo "t=$(( $(date +%s) - T0 ))s"
python - <<PY
import json
for f in ("$OUT/${1}_bench.json", "$OUT/${1}_bench_driver_style.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value", d["value"], "parity", (d.get("value_parity") or {}).get("value"), "geom", (d.get("with_geometry") or {}).get("value"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
        print(" roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "frac_algorithmic", "launch_ms", "tiles_left_out")}, d["roofline"]["rocprof_trace_note"][-30:])
        rp = d.get("roofline_parity")
        if rp: print(" roofline_parity", {k: rp[k] for k in ("achieved", "frac", "frac_algorithmic", "launch_ms", "tiles_left_out")}, rp["superpoint_ms_per_keyframe"])
    except Exception as e:
        print(f, "parse failed", e)
PY
head -12 $OUT/${1}f_kernel_stats.md | cut -c1-200
head -12 $OUT/${1}s_kernel_stats.md | cut -c1-200
