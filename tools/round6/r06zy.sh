#!/bin/bash
# round 6 evidence at the last kernel change: rocprofv3 kernel trace + four PMC passes, fp16 (r06zf) and split (r06zs) (tools/profile_gpu.sh)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
T0=$(date +%s)
bash tools/profile_gpu.sh r06zs --precision split > $OUT/r06zs_run.log 2>&1; echo "split profile rc=$?"
bash tools/profile_gpu.sh r06zf > $OUT/r06zf_run.log 2>&1; echo "f16 profile rc=$?"
echo "t=$(( $(date +%s) - T0 ))s"
head -24 $OUT/r06zs_kernel_stats.md | cut -c1-220
cat $OUT/r06zs_traffic.json | head -80
grep -i "wino" $OUT/r06zs_pmc_summary.md | head -20 | cut -c1-400
