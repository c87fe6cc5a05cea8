#!/bin/bash
# round 4: the detector step collected one unit later -- same decisions, and does the GPU stay busy now?
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_e2e_depth.py tests/test_gpu_e2e_scene.py tests/test_gpu_bench_shape.py -x -q -m gpu -k "pinhole_depth or rendered_scene or cpp_host_loop" > gpurun_out/r04w_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r04w_pytest.log
B="--no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --c5-rows 0"
timeout 600 python bench.py $B > gpurun_out/r04w_bench_async.json 2> gpurun_out/r04w_bench_async.err; echo "async rc=$?"
OMNI_DETECTOR_ASYNC=0 timeout 600 python bench.py $B > gpurun_out/r04w_bench_sync.json 2> gpurun_out/r04w_bench_sync.err; echo "sync rc=$?"
python - <<'PY'
import json
for t in ("async","sync"):
    d=json.load(open(f'gpurun_out/r04w_bench_{t}.json'))
    print(t,'value',d['value'],'ms',d['ms_per_step'],'parity',d['value_parity']['value'], 'lat p50', d['keyframe_latency_ms']['p50'], d['value_parity']['keyframe_latency_ms']['p50'])
    print('   host f16  ', d.get('host_ms_per_microbatch'))
    print('   host split', d['value_parity'].get('host_ms_per_microbatch'))
PY
