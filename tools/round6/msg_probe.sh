# a unit's key-frame messages filled by OMNI_MESSAGE_THREADS helper threads + the caller (0 = inline) at 20 / 24 / 200 key frames per region, interleaved; the host-loop tests
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_cpp_host.py tests/test_gpu_bench_shape.py -q -x -m gpu 2>&1 | grep -E "passed|failed" | tail -2
for r in 1 2; do for s in 20 24 200; do for t in 0 3; do
OMNI_MESSAGE_THREADS=$t python bench.py --steps $s --warmup 5 --no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --parity-steps 0 --c5-rows 0 --long-region-steps 0 --batched-rows 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('steps', $s, 'msg_threads', $t, 'value', d['value'], 'ms/step', d['ms_per_step'], d.get('host_ms_per_microbatch'))"
done; done; done
} 2>&1 | tee gpurun_out/r06l_message_threads.log
