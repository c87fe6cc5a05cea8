#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
LEGS="--no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --parity-steps 0 --c5-rows 0 --batched-rows 0 --match-db-rows 1000"
for F in 0 1; do
OMNI_SPLIT_FUSE1A=$F OMNI_SPLIT_TRACE=1 OMNI_LIB=omni-swarm_amd/lib_abl/libomni_hip_abltrace.so timeout 300 python bench.py --precision split --steps 8 --warmup 8 --min-time 0 $LEGS > $OUT/r04h_trace$F.json 2> $OUT/r04h_trace$F.err
grep "split step trace\|split trace" $OUT/r04h_trace$F.err | grep "wave 0" | head -40
done
