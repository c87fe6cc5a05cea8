#!/bin/bash
# round 4, GPU call c: conv1a built inside the split conv1b kernel (FUSE1A): parity tests, A/B against the separate conv1a pass, in-kernel trace
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TAG=${1:-r04c}
LEGS="--no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --parity-steps 0 --c5-rows 0 --batched-rows 0"
timeout 900 python -m pytest tests/test_gpu_superpoint.py tests/test_gpu_mask_skip.py tests/test_gpu_bench_shape.py -m gpu -q -x -k "f32_layers or batch_equals or mask_skip or split_precision" > $OUT/${TAG}_pytest.log 2>&1
echo "pytest rc=$?"; tail -15 $OUT/${TAG}_pytest.log
OMNI_SPLIT_FZ_MIX=0 timeout 600 python -m pytest tests/test_gpu_superpoint.py -m gpu -q -x -k "f32_layers and SPLIT and not UNFUSED" > $OUT/${TAG}_pytest_nomix.log 2>&1
echo "pytest nomix rc=$?"; tail -3 $OUT/${TAG}_pytest_nomix.log
for V in "0 1" "1 1" "1 0"; do
  set -- $V
  OMNI_SPLIT_FUSE1A=$1 OMNI_SPLIT_FZ_MIX=$2 timeout 300 python bench.py --precision split --steps 64 --warmup 16 $LEGS > $OUT/${TAG}_bench_fuse$1_mix$2.json 2> $OUT/${TAG}_bench_fuse$1_mix$2.err
  echo "fuse=$1 mix=$2 rc=$?"; python - <<PY
import json
try:
    d = json.loads(open("$OUT/${TAG}_bench_fuse$1_mix$2.json").read().strip().splitlines()[-1])
    print("  value", d["value"], "ms", d["ms_per_step"], "stages", d["roofline"]["stages_ms_per_keyframe"])
except Exception as e:
    print("  parse failed", e)
PY
done
OMNI_SPLIT_TRACE=1 timeout 300 python bench.py --precision split --steps 16 --warmup 8 --min-time 0 $LEGS > $OUT/${TAG}_split_trace.json 2> $OUT/${TAG}_split_trace.err
grep "split trace" $OUT/${TAG}_split_trace.err | grep "fuse1a=1" | head -16
