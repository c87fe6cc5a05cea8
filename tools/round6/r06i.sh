#!/bin/bash
# fp16 conv1a inside conv1b's kernel without the u8 -> (hi, lo) table (OMNI_PP_U8=1: operands straight from the bytes) against the table (=0):
# the fp16 GPU tests, then stage times interleaved
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_superpoint.py tests/test_gpu_mask_skip.py tests/test_gpu_bench_shape.py -q -x -m gpu -k "f16 or F16 or bit_identical or mask or headline" 2>&1 | tail -4
B="python bench.py --no-cpu-baseline --steps 64 --warmup 16 --match-db-rows 8192 --batched-rows 0 --big-db-keyframes 0 --f32-steps 0 --c5-rows 0 --parity-steps 0 --geometry-steps 0 --python-steps 0 --long-region-steps 0"
for r in 1 2 3; do for m in 0 1; do
  OMNI_PP_U8=$m timeout 300 $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['stages_ms_per_keyframe']
print('pp_u8=$m', 'value', d['value'], 'conv1b', s['conv1b+pool'], 'frac', r['frac'], 'launch_ms', r.get('launch_ms'), 'sp_ms/kf', r.get('superpoint_ms_per_keyframe'), 'parity', d.get('parity'))"
done; done
} 2>&1 | tee gpurun_out/r06i_pp_u8.log
