#!/bin/bash
# run() recutting a 20-key-frame region into equal units: decisions (pytest) and the driver-style line for the three plans, interleaved
python -m pytest tests/test_gpu_bench_shape.py -q -x -m gpu -k "recuts or cpp_host_loop or streaming" 2>&1 | tail -3
mkdir -p gpurun_out
for r in 1 2; do for p in 0 1 2; do
  OMNI_PIPELINE_UNIT_PLAN=$p timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --big-db-keyframes 0 --f32-steps 0 --c5-rows 0 --parity-steps 0 --geometry-steps 0 --python-steps 0 --match-db-rows 8192 --batched-rows 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('plan=$p value', d['value'], 'ms_per_step', d['ms_per_step'], 'long', d.get('value_long_regions'))"
done; done 2>&1 | tee gpurun_out/r06c_unit_plan.log
