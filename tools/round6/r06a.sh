#!/bin/bash
# Winograd split kernels, first correct version: stage times of OMNI_PREC_SPLIT with OMNI_SPLIT_WINO=0 / 7 on the same box, interleaved
mkdir -p gpurun_out
B="python bench.py --precision split --no-cpu-baseline --steps 64 --warmup 16 --match-db-rows 8192 --batched-rows 0 --big-db-keyframes 0 --f32-steps 0 --c5-rows 0 --parity-steps 0 --geometry-steps 0 --python-steps 0 --long-region-steps 0"
for r in 1 2; do
  for m in 0 7; do
    OMNI_SPLIT_WINO=$m timeout 300 $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['stages_ms_per_keyframe']
print('wino=$m', 'value', d['value'], ' '.join(f'{k}={v}' for k,v in s.items()), 'sp_ms/kf', r.get('superpoint_ms_per_keyframe'))"
  done
done 2>&1 | tee gpurun_out/r06a_stage_ab_wino.log
