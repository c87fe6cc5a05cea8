#!/bin/bash
mkdir -p gpurun_out
B="--no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --c5-rows 0"
for V in "OMNI_PIPELINE_ONE_STREAM=1 PIPES=3" "OMNI_PIPELINE_ONE_STREAM=0 PIPES=3" "OMNI_PIPELINE_ONE_STREAM=0 PIPES=4" "OMNI_PIPELINE_ONE_STREAM=1 PIPES=4"; do
  P=2; case "$V" in *PIPES=3*) P=3;; *PIPES=4*) P=4;; esac
  env $V timeout 600 python bench.py $B --pipelines $P > gpurun_out/r04x_bench.json 2> gpurun_out/r04x_bench.err; echo "$V rc=$?"
  python - <<'PY'
import json
d=json.load(open('gpurun_out/r04x_bench.json'))
print('   value',d['value'],'ms',d['ms_per_step'],'parity',d['value_parity']['value'], 'lat p50', d['keyframe_latency_ms']['p50'], d['value_parity']['keyframe_latency_ms']['p50'])
print('   host f16  ', d.get('host_ms_per_microbatch'))
PY
done
