#!/bin/bash
# conv3a (64 -> 128) as a Winograd layer (OMNI_SPLIT_WINO bit 3): per-layer errors for every mask, the split GPU tests, stage times 7 vs 15 interleaved
mkdir -p gpurun_out
{
timeout 900 python tools/round6/wino_check.py 2>&1 | grep -v "^ *$" | tail -70
timeout 1200 python -m pytest tests/test_gpu_superpoint.py tests/test_gpu_mask_skip.py tests/test_gpu_bench_shape.py -q -x -m gpu -k "split or SPLIT or mask or winograd" 2>&1 | tail -4
B="python bench.py --precision split --no-cpu-baseline --steps 64 --warmup 16 --match-db-rows 8192 --batched-rows 0 --big-db-keyframes 0 --f32-steps 0 --c5-rows 0 --parity-steps 0 --geometry-steps 0 --python-steps 0 --long-region-steps 0"
for r in 1 2; do for m in 7 15; do
  OMNI_SPLIT_WINO=$m timeout 300 $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['stages_ms_per_keyframe']
print('wino=$m', 'value', d['value'], ' '.join(f'{k}={v}' for k,v in s.items() if k.startswith('conv')), 'sp_ms/kf', r.get('superpoint_ms_per_keyframe'))"
done; done
} 2>&1 | tee gpurun_out/r06g_conv3a_wino.log
