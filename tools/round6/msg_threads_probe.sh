for r in 1 2; do for t in 3 6; do OMNI_MESSAGE_THREADS=$t python bench.py --steps 20 --warmup 5 --no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --parity-steps 0 --c5-rows 0 --long-region-steps 0 --batched-rows 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('msg_threads', $t, 'value', d['value'], 'ms/step', d['ms_per_step'], d.get('host_ms_per_microbatch'))"; done; done
