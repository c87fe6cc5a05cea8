#!/bin/bash
# last check of the bench line at HEAD (driver style): valid JSON, the fields the docs quote
mkdir -p gpurun_out
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04ac_bench_driver_style.json 2> gpurun_out/r04ac_bench.err; echo "rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04ac_bench_driver_style.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','higher_is_better','scaling','vs_baseline','dtype','data')})
print(d['config'])
print('long', (d.get('value_long_regions') or {}).get('value'), 'parity', d['value_parity']['value'], 'roofline', d['roofline']['frac'], d['roofline']['bound'], d['roofline']['traffic'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['kind'])
PY
