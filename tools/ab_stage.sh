#!/bin/bash
# Cheap same-box A/B of library builds on kernel stage times only (no index builds, no detector): ~10 s per variant per round.
#   tools/ab_stage.sh <rounds> <variant suffixes...>      "cur" = omni-swarm_amd/lib/libomni_hip.so, X = libomni_hip_X.so
# Prints SuperPoint per-stage ms per key frame at the bench's launch shape (BATCH, default 64 images) and MobileNetVLAD stand-alone.
R=${1:-2}; shift
export BATCH=${BATCH:-64}
for r in $(seq 1 $R); do
  for v in "$@"; do
    if [ $v = cur ]; then unset OMNI_LIB; else export OMNI_LIB=$PWD/omni-swarm_amd/lib/libomni_hip_$v.so; fi
    echo "== $v (round $r)"; timeout 120 python tools/stage_timing.py 2>&1 | tail -2
  done
done
