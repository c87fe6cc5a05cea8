#!/bin/bash
# round 4: where does the host thread's time go per micro-batch?
mkdir -p gpurun_out
B="--no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --c5-rows 0"
timeout 600 python bench.py $B > gpurun_out/r04u_bench.json 2> gpurun_out/r04u_bench.err; echo "rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04u_bench.json'))
print('value',d['value'],'ms',d['ms_per_step'],'parity',d['value_parity']['value'])
print('host f16  ', d.get('host_ms_per_microbatch'))
print('host split', d['value_parity'].get('host_ms_per_microbatch'))
PY
