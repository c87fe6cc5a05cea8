#!/bin/bash
# conv1b (fused conv1a + conv1b ping-pong kernel) phase anatomy on the GPU box: the s_memtime trace of the service phase and the timing
# ablations of OMNI_PP_DBG (bit 0: no tile build, bit 1: no epilogue; WRONG results, timing only).  BATCH images per launch.
export BATCH=${BATCH:-64}
for dbg in 0 1 2 3; do
  echo "== OMNI_PP_DBG=$dbg"
  OMNI_PP_DBG=$dbg timeout 120 python tools/stage_timing.py 2>&1 | grep "SuperPoint batch" | sed 's/.*{\(.*\)}.*/\1/' | tr ',' '\n' | grep -E "conv1b|conv2a|conv2b"
done
echo "== trace"
OMNI_PP_TRACE=1 timeout 120 python tools/stage_timing.py 2>&1 | grep "pp trace" | head -8
