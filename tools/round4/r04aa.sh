#!/bin/bash
# round 4: units in flight oldest-first (OMNI_PIPELINE_FIFO) at the default region (200 key frames) and at the driver's (20)
mkdir -p gpurun_out
B="--no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --c5-rows 0"
run() {  # label env... -- args
  L=$1; shift
  env "$@" > /dev/null 2>&1
}
for CFG in "FIFO=0 P=0" "FIFO=1 P=0" "FIFO=2 P=0" "FIFO=1 P=2" "FIFO=1 P=3"; do
  eval $CFG
  for K in "--steps 200 --warmup 20" "--steps 20 --warmup 5"; do
    OMNI_PIPELINE_FIFO=$FIFO timeout 600 python bench.py $B $K --pipelines $P > gpurun_out/r04aa_bench.json 2> gpurun_out/r04aa_bench.err
    python - "$CFG" "$K" <<'PY'
import json,sys
d=json.load(open('gpurun_out/r04aa_bench.json'))
print(sys.argv[1], '|', sys.argv[2], '| value',d['value'],'parity',d['value_parity']['value'], 'lat p50', d['keyframe_latency_ms']['p50'], d['value_parity']['keyframe_latency_ms']['p50'], 'wait_gpu', d['host_ms_per_microbatch']['wait_gpu'])
PY
  done
done
