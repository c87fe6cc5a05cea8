#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
LEGS="--no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --parity-steps 0 --c5-rows 0 --batched-rows 0 --match-db-rows 1000"
for N in 0 $@; do
  LIB=omni-swarm_amd/lib_abl/libomni_hip_abl$N.so; [ $N = 0 ] && LIB=omni-swarm_amd/lib/libomni_hip.so
  OMNI_LIB=$LIB timeout 300 python bench.py --precision split --steps 16 --warmup 8 --min-time 0 $LEGS > $OUT/r04e_abl$N.json 2> $OUT/r04e_abl$N.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/r04e_abl$N.json").read().strip().splitlines()[-1])
    print("abl $N conv1b", d["roofline"]["stages_ms_per_keyframe"]["conv1b+pool"], "launch_ms", d["roofline"]["launch_ms"])
except Exception as e:
    print("abl $N parse failed", e)
PY
done
