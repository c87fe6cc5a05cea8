#!/bin/bash
# round 5, call 11: s_memtime trace of the fused fp16 conv1a + conv1b kernel's service phase (OMNI_PP_TRACE=1) and of the unfused layers
set -u
export TMPDIR=/tmp
PREC=f16 BATCH=64 NO_VLAD=1 OMNI_SP_PROFILE_MASK=1 OMNI_PP_TRACE=1 timeout 200 python tools/stage_timing.py 2>&1 | grep -E "pp trace|SuperPoint" | head -40
