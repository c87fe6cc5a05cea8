#!/bin/bash
# round 5, call 6: A/B of the split block kernel's register budget for the cin <= 16 shapes (lib/libomni_hip_w3.so: 3 waves per SIMD, some scratch)
set -u
OUT=gpurun_out
export TMPDIR=/tmp
for V in cur w3 cur w3; do
  if [ $V = cur ]; then unset OMNI_LIB; else export OMNI_LIB=$PWD/omni-swarm_amd/lib/libomni_hip_$V.so; fi
  echo "== $V"; bash tools/vlad_seq_trace.sh f32 32 2>&1 | grep -E "sblock|stem|sum" | cut -c1-110
  BATCH=32 PREC=f32 timeout 200 python tools/vlad_trace32.py 2>&1 | tail -1
done > $OUT/r05f_vlad_wpe_ab.log 2>&1
cat $OUT/r05f_vlad_wpe_ab.log
