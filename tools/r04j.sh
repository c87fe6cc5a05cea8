#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TAG=${1:-r04j}
LEGS="--no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --parity-steps 0 --c5-rows 0 --batched-rows 0 --match-db-rows 1000"
timeout 900 python -m pytest tests/test_gpu_superpoint.py tests/test_gpu_bench_shape.py -m gpu -q -x -k "f32_layers or batch_equals or split_precision" > $OUT/${TAG}_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/${TAG}_pytest.log
for F in 1; do
OMNI_SPLIT_FUSE1A=$F timeout 300 python bench.py --precision split --steps 64 --warmup 16 $LEGS > $OUT/${TAG}_bench_fuse$F.json 2> $OUT/${TAG}_bench_fuse$F.err
python - <<PY
import json
try:
    d = json.loads(open("$OUT/${TAG}_bench_fuse$F.json").read().strip().splitlines()[-1])
    print("fuse $F value", d["value"], "ms", d["ms_per_step"], "stages", d["roofline"]["stages_ms_per_keyframe"])
except Exception as e:
    print("  parse failed", e)
PY
done
OMNI_SPLIT_TRACE=1 OMNI_LIB=omni-swarm_amd/lib_abl/libomni_hip_abltrace.so timeout 300 python bench.py --precision split --steps 8 --warmup 8 --min-time 0 $LEGS > $OUT/${TAG}_trace.json 2> $OUT/${TAG}_trace.err
grep "step trace" $OUT/${TAG}_trace.err | grep "wave 0 tile 4" | awk '!seen[$0]++' | head -12
grep "split trace" $OUT/${TAG}_trace.err | grep "wave 0 tile 4" | awk '!seen[$0]++' | head -12
