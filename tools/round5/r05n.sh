#!/bin/bash
# round 5: the messages' heavy part behind the detector enqueue -- host-loop tests + the driver-style line
set -u
OUT=gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bench_shape.py tests/test_gpu_e2e_scene.py tests/test_gpu_e2e_depth.py tests/test_gpu_shard_rccl.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error" | tail -3
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --c5-rows 0 --batched-rows 0 --match-db-rows 4096 > $OUT/r05n_bench.json 2> $OUT/r05n.err
python -c "
import json; d = json.loads(open('$OUT/r05n_bench.json').read().strip().splitlines()[-1]); print('value', d['value'], 'long', d['value_long_regions']['value'], 'parity', d['value_parity']['value'], d['host_ms_per_microbatch'], d['keyframe_latency_ms']['p50'])"
done
