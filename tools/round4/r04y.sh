#!/bin/bash
# round 4 evidence, part 1: kernel trace + 4 PMC passes at HEAD, f16 and split (summaries -> profiles/r04y*)
set -u
export TMPDIR=/tmp
T0=$(date +%s)
bash tools/profile_gpu.sh r04yf
echo "t=$(( $(date +%s) - T0 ))s"
bash tools/profile_gpu.sh r04ys --precision split
echo "t=$(( $(date +%s) - T0 ))s"
head -14 gpurun_out/r04yf_kernel_stats.md | cut -c1-220
head -14 gpurun_out/r04ys_kernel_stats.md | cut -c1-220
cat gpurun_out/r04yf_traffic.json | head -30
cat gpurun_out/r04ys_traffic.json | head -30
