#!/bin/bash
# rocprofv3 kernel trace of the 64-query fp16 search (tools/knn_batch_timing.py): tools/knn_trace.sh <tag> [ENV=VAL ...]
TAG=${1:-knn}; shift
for kv in "$@"; do export "$kv"; done
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/${TAG}_trace -o ${TAG} -- python $REPO/tools/knn_batch_timing.py > $REPO/gpurun_out/${TAG}.log 2>&1
DB=$(find $REPO/gpurun_out/${TAG}_trace -name "*_results.db" | head -1)
python $REPO/tools/rocprof_summary.py $DB "64-query fp16 search (${TAG})" | head -16
tail -3 $REPO/gpurun_out/${TAG}.log
