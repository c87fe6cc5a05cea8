#!/bin/bash
# round 5: FULLY SERIAL kernel traces (one unit in flight, MobileNetVLAD behind SuperPoint on one stream: OMNI_PIPELINE_ONE_STREAM=1): every kernel's duration
# is its stand-alone time -- the per-stage figures of DESIGN.md can be checked against these medians
set -u
OUT=gpurun_out
export TMPDIR=/tmp
LEGS="--no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --parity-steps 0 --c5-rows 0 --long-region-steps 0 --batched-rows 0 --match-db-rows 4096"
for P in f16 split; do
  TAG=r05p_${P}_serial
  rm -rf $OUT/${TAG}_trace
  OMNI_PIPELINE_ONE_STREAM=1 timeout 300 rocprofv3 --kernel-trace -d $OUT/${TAG}_trace -o ${TAG} -- python bench.py --precision $P --pipelines 1 --steps 40 --warmup 8 --min-time 0 $LEGS > $OUT/${TAG}_bench.json 2> $OUT/${TAG}.err
  python tools/rocprof_summary.py $(ls $OUT/${TAG}_trace/*_results.db $OUT/${TAG}_trace/*/*_results.db 2>/dev/null | head -1) --json $OUT/${TAG}_kernel_times_su.json "(${TAG}: OMNI_PIPELINE_ONE_STREAM=1 bench.py --precision $P --pipelines 1 --steps 40: one unit in flight on ONE stream, no two kernels overlap; 1x MI355X)" > $OUT/${TAG}_kernel_stats.md 2>> $OUT/${TAG}.err
  rm -rf $OUT/${TAG}_trace
  python -c "
import json; d = json.loads(open('$OUT/${TAG}_bench.json').read().strip().splitlines()[-1]); print('$P serial value', d['value'], 'ms/unit', d['ms_per_step'] * 8)
t = json.load(open('$OUT/${TAG}_kernel_times_su.json'))['kernels']
tot = 0
for k, v in t.items():
    if v['calls'] >= 5 and not k.startswith('__amd') and 'ip_scan' not in k and 'row_norm' not in k and 'f32_to_t16' not in k:
        per_unit = v['median_us'] * (v['calls'] / 5.0 if v['calls'] % 5 == 0 else v['calls'] / 5.0)
print(sum(v['median_us'] * v['calls'] for k, v in t.items() if not k.startswith('__amd') and 'ip_scan_kernel' not in k and 'row_norm' not in k and 'f32_to_t16' not in k) / 1e3, 'ms of kernels in the run')
"
  head -34 $OUT/${TAG}_kernel_stats.md | cut -c1-180 | tail -28
done
