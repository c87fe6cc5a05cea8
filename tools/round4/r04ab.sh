#!/bin/bash
# round 4: the driver's 20-key-frame regions -- units in flight x oldest-first
mkdir -p gpurun_out
B="--no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --c5-rows 0 --parity-steps 0 --long-region-steps 0 --steps 20 --warmup 5"
for CFG in "FIFO=0 P=2" "FIFO=0 P=3" "FIFO=1 P=2" "FIFO=1 P=3" "FIFO=2 P=3" "FIFO=0 P=3" "FIFO=1 P=3"; do
  eval $CFG
  OMNI_PIPELINE_FIFO=$FIFO timeout 600 python bench.py $B --pipelines $P > gpurun_out/r04ab_bench.json 2> gpurun_out/r04ab_bench.err
  python - "$CFG" <<'PY'
import json,sys
d=json.load(open('gpurun_out/r04ab_bench.json'))
print(sys.argv[1], '| value',d['value'], 'repeats', d['repeats'], 'minmax', d['ms_per_step_minmax'], 'lat p50', d['keyframe_latency_ms']['p50'], 'wait_gpu', d['host_ms_per_microbatch']['wait_gpu'])
PY
done
