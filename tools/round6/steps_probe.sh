for s in 20 24 16 40 200; do
python bench.py --steps $s --warmup 5 --no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --parity-steps 0 --c5-rows 0 --long-region-steps 0 --batched-rows 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('steps', $s, 'value', d['value'], 'ms/step', d['ms_per_step'], d.get('host_ms_per_microbatch'))"
done
