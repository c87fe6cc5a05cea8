#!/bin/bash
# where a 20-key-frame region goes: kernel trace of the driver-style headline leg, span / busy / idle gaps per region (tools/region_timeline.py)
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/r06j_trace
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/r06j_trace -o r06j -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --parity-steps 0 --c5-rows 0 --long-region-steps 0 --batched-rows 0 > gpurun_out/r06j_bench.json 2> gpurun_out/r06j_err.log
DB=$(ls gpurun_out/r06j_trace/*_results.db gpurun_out/r06j_trace/*/*_results.db 2>/dev/null | head -1)
python tools/region_timeline.py $DB 300 2>&1 | tee gpurun_out/r06j_region_timeline.log
python - "$DB" <<'PY' 2>&1 | tee -a gpurun_out/r06j_region_timeline.log
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'copy' in t.lower() or 'memory' in t.lower()][:10])
PY
tail -1 gpurun_out/r06j_bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms_per_step', d['ms_per_step'], d.get('host_ms_per_microbatch'))"
find gpurun_out/r06j_trace -name '*.db' -size +20M -delete
