# the oldest-first rule of KeyframePipeline::run (OMNI_PIPELINE_FIFO: 0 / 1 = no chain / the SuperPoint streams chained) by units per run() call (8 key frames each)
for s in ${STEPS:-24 32 40 48 56 64}; do for f in 0 1; do
OMNI_PIPELINE_FIFO=$f python bench.py --steps $s --warmup 8 --no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --parity-steps 0 --c5-rows 0 --long-region-steps 0 --batched-rows 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('steps', $s, 'fifo', $f, 'value', d['value'], 'ms/step', d['ms_per_step'], 'minmax', d['ms_per_step_minmax'])"
done; done 2>&1 | tee gpurun_out/r06k_fifo_by_units.log
