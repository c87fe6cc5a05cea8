#!/bin/bash
# MobileNetVLAD alone at the bench's launch shape (32 images): rocprofv3 kernel trace summary + host-clock time.  Run on the GPU box:
#   tools/vlad_ab.sh <tag>     -> gpurun_out/<tag>_vlad32_kernel_stats.md
set -e
TAG=${1:-vlad}
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
python $REPO/tools/vlad_trace32.py
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/${TAG}_trace -o ${TAG} -- python $REPO/tools/vlad_trace32.py > $REPO/gpurun_out/${TAG}_trace.log 2>&1 || true
DB=$(find $REPO/gpurun_out/${TAG}_trace -name "*_results.db" | head -1)
python $REPO/tools/rocprof_summary.py $DB "MobileNetVLAD alone, 32 images of 600x480 per launch sequence (${TAG})" > $REPO/gpurun_out/${TAG}_vlad32_kernel_stats.md || true
head -30 $REPO/gpurun_out/${TAG}_vlad32_kernel_stats.md
