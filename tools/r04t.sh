#!/bin/bash
# round 4: hardware queues + a high-priority detector stream: does the pipeline overlap now?
mkdir -p gpurun_out
B="--no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --c5-rows 0"
timeout 600 python bench.py $B > gpurun_out/r04t_bench_q8.json 2> gpurun_out/r04t_bench_q8.err; echo "q8 rc=$?"
OMNI_HW_QUEUES=0 timeout 600 python bench.py $B > gpurun_out/r04t_bench_q4.json 2> gpurun_out/r04t_bench_q4.err; echo "q4 rc=$?"
python - <<'PY'
import json
for t in ("q8","q4"):
    d=json.load(open(f'gpurun_out/r04t_bench_{t}.json'))
    print(t,'value',d['value'],'ms',d['ms_per_step'],'parity',d['value_parity']['value'], 'lat p50', d['keyframe_latency_ms']['p50'])
PY
