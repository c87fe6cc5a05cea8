#!/bin/bash
# round 4: the upload in two halves (MobileNetVLAD starts on the up cameras' half) -- cam tests + the driver-style line twice
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bench_shape.py -x -q -m gpu -k "cam_unit or row_stride or cpp_host_loop" > gpurun_out/r04af_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r04af_pytest.log
B="--no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --c5-rows 0 --parity-steps 0"
for i in 1 2; do
timeout 600 python bench.py $B --steps 20 --warmup 5 > gpurun_out/r04af_bench.json 2> gpurun_out/r04af_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04af_bench.json'))
print('value',d['value'],'long',(d.get('value_long_regions') or {}).get('value'), d['ms_per_step_minmax'])
PY
done
