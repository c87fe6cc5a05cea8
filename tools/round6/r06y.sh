#!/bin/bash
# round 6 evidence: rocprofv3 kernel trace + four PMC passes at HEAD, fp16 and split (tools/profile_gpu.sh)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
T0=$(date +%s)
bash tools/profile_gpu.sh r06ys --precision split > $OUT/r06ys_run.log 2>&1; echo "split profile rc=$?"
bash tools/profile_gpu.sh r06yf > $OUT/r06yf_run.log 2>&1; echo "f16 profile rc=$?"
echo "t=$(( $(date +%s) - T0 ))s"
head -24 $OUT/r06ys_kernel_stats.md | cut -c1-220
cat $OUT/r06ys_traffic.json | head -80
grep -i "wino" $OUT/r06ys_pmc_summary.md | head -20 | cut -c1-400
