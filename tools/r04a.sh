#!/bin/bash
# round 4, GPU call a: the split-precision mask skip (merged from wip/split-mask-skip) validated and measured; headline on this box
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
LEGS="--no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --parity-steps 0 --c5-rows 0 --batched-rows 0"
timeout 600 python -m pytest tests/test_gpu_mask_skip.py -m gpu -q -x > $OUT/r04a_pytest_mask_skip.log 2>&1
echo "mask-skip test rc=$?" 
tail -3 $OUT/r04a_pytest_mask_skip.log
for F in 0 1; do
  OMNI_SP_MASK_SKIP_SPLIT=$F timeout 300 python bench.py --precision split --steps 64 --warmup 16 $LEGS > $OUT/r04a_bench_split_skip$F.json 2> $OUT/r04a_bench_split_skip$F.err
  echo "split skip=$F rc=$?"; python - <<PY
import json
try:
    d = json.loads(open("$OUT/r04a_bench_split_skip$F.json").read().strip().splitlines()[-1])
    print("  value", d["value"], "ms", d["ms_per_step"], "stages", d["roofline"]["stages_ms_per_keyframe"])
except Exception as e:
    print("  parse failed", e)
PY
done
OMNI_SPLIT_TRACE=1 OMNI_SP_MASK_SKIP_SPLIT=1 timeout 300 python bench.py --precision split --steps 16 --warmup 8 --min-time 0 $LEGS > $OUT/r04a_split_trace.json 2> $OUT/r04a_split_trace.err
grep "split trace" $OUT/r04a_split_trace.err | head -24
timeout 600 python bench.py > $OUT/r04a_bench_default.json 2> $OUT/r04a_bench_default.err
echo "default bench rc=$?"
python - <<PY
import json
d = json.loads(open("$OUT/r04a_bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "parity", d["value_parity"]["value"] if d.get("value_parity") else None, "geom", d["with_geometry"]["value"] if d.get("with_geometry") else None)
print("stages", d["roofline"]["stages_ms_per_keyframe"])
PY
