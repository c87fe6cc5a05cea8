#!/bin/bash
# round 4: the mask's constant region left out of conv3b's tile walk on the fp16 register-stationary kernel too
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_mask_skip.py tests/test_gpu_superpoint.py tests/test_gpu_bench_shape.py -x -q -m gpu -k "not persistent_kernels and not split_precision and not cpp_host_loop" > gpurun_out/r04ae_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r04ae_pytest.log
B="--no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --c5-rows 0 --parity-steps 0"
timeout 600 python bench.py $B --steps 20 --warmup 5 > gpurun_out/r04ae_bench.json 2> gpurun_out/r04ae_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04ae_bench.json'))
print('value',d['value'],'long',(d.get('value_long_regions') or {}).get('value'))
print(d['roofline']['stages_ms_per_keyframe'])
print(d['roofline']['mask_skip']['tiles_left_out'])
PY
