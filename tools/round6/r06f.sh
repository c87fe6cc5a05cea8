#!/bin/bash
# fp16 cin = 128 layers: the v5 register-stationary kernel (OMNI_CONV_RS=2: epilogue + DMA inside the stream) against v4 (=1): bit identity, stage times
mkdir -p gpurun_out
OMNI_CONV_RS=2 python -m pytest tests/test_gpu_superpoint.py tests/test_gpu_mask_skip.py -q -x -m gpu -k "f16 or bit_identical or mask" 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --steps 64 --warmup 16 --match-db-rows 8192 --batched-rows 0 --big-db-keyframes 0 --f32-steps 0 --c5-rows 0 --parity-steps 0 --geometry-steps 0 --python-steps 0 --long-region-steps 0"
for r in 1 2; do for m in 1 2; do
  OMNI_CONV_RS=$m timeout 300 $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['stages_ms_per_keyframe']
print('rs=$m', 'value', d['value'], ' '.join(f'{k}={v}' for k,v in s.items() if k.startswith('conv3') or k.startswith('conv4') or k.startswith('convPa')), 'sp_ms/kf', r.get('superpoint_ms_per_keyframe'))"
done; done 2>&1 | tee gpurun_out/r06f_stage_ab_rs2.log
