#!/bin/bash
# round 4: sparse split convDa -- bit identity with the dense path, the north-star bar, then the parity bench
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_superpoint.py -x -q -m gpu -k "sparse or f32_layers" > gpurun_out/r04p_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r04p_pytest.log
timeout 600 python -m pytest tests/test_gpu_bench_shape.py -x -q -m gpu -k "split_precision" > gpurun_out/r04p_pytest2.log 2>&1
echo "pytest2 rc=$?"; tail -5 gpurun_out/r04p_pytest2.log
timeout 600 python bench.py --no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --c5-rows 0 > gpurun_out/r04p_bench.json 2> gpurun_out/r04p_bench.err
echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04p_bench.json'))
print('value',d['value'],'parity',d['value_parity']['value'])
print(json.dumps(d['roofline_parity']['stages_ms_per_keyframe']))
PY
