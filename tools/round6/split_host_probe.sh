# the split (tolerance-meeting) mode's host-loop settings re-probed with the Winograd kernels: units in flight x the oldest-first chain, 64 key frames per region
mkdir -p gpurun_out
{
for r in 1 2; do for c in "0 -1" "2 0" "2 2" "3 -1" "3 0" "4 -1" "4 0"; do set -- $c
OMNI_PIPELINE_FIFO=$2 python bench.py --precision split --pipelines $1 --steps 64 --warmup 16 --no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --parity-steps 0 --c5-rows 0 --long-region-steps 0 --batched-rows 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pipelines', $1, 'fifo', $2, 'value', d['value'], 'ms/step', d['ms_per_step'], d.get('host_ms_per_microbatch'))"
done; done
} 2>&1 | tee gpurun_out/r06m_split_host_probe.log
