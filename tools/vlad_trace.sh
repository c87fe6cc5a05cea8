#!/bin/bash
# per-kernel stand-alone MobileNetVLAD times at BATCH images (half of them go through the net): tools/vlad_trace.sh <batch>
export TMPDIR=/tmp BATCH=${1:-32}
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/vt -o vt -- python tools/stage_timing.py > /dev/null 2>&1
python tools/rocprof_summary.py gpurun_out/vt/vt_results.db "(vlad stand-alone, BATCH=$BATCH)" | grep -E "vlad|calls" | cut -c1-125
rm -rf gpurun_out/vt
