#!/bin/bash
# round 4: sp_cand_kernel with one thread per candidate -- bit-exact post-processing, stage time
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sp_post.py tests/test_gpu_superpoint.py -x -q -m gpu -k "not persistent_kernels" > gpurun_out/r04ad_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r04ad_pytest.log
B="--no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --c5-rows 0"
timeout 600 python bench.py $B --steps 20 --warmup 5 > gpurun_out/r04ad_bench.json 2> gpurun_out/r04ad_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04ad_bench.json'))
print('value',d['value'],'long',(d.get('value_long_regions') or {}).get('value'),'parity',d['value_parity']['value'])
print(d['roofline']['stages_ms_per_keyframe'])
PY
