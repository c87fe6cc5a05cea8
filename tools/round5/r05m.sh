#!/bin/bash
# round 5: where a 20-key-frame region loses against the steady state -- kernel timeline per region (tools/region_timeline.py)
set -u
OUT=gpurun_out
export TMPDIR=/tmp
LEGS="--no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --parity-steps 0 --c5-rows 0 --long-region-steps 0 --batched-rows 0 --match-db-rows 4096"
rm -rf $OUT/r05m_trace
timeout 300 rocprofv3 --kernel-trace -d $OUT/r05m_trace -o r05m -- python bench.py --steps 20 --warmup 5 --min-time 0.3 $LEGS > $OUT/r05m_bench.json 2> $OUT/r05m.err
python -c "
import json; d = json.loads(open('$OUT/r05m_bench.json').read().strip().splitlines()[-1]); print('value', d['value'], 'ms/region', d['ms_per_step'] * 20, d['host_ms_per_microbatch'])"
python tools/region_timeline.py $(ls $OUT/r05m_trace/*_results.db $OUT/r05m_trace/*/*_results.db 2>/dev/null | head -1) 300
rm -rf $OUT/r05m_trace
