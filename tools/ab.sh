#!/bin/bash
# A/B several builds of libomni_hip.so on the same GPU box, interleaved: tools/ab.sh <rounds> <variant suffixes...>  ("cur" = the default build)
R=${1:-2}; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    if [ $v = cur ]; then unset OMNI_LIB; else export OMNI_LIB=$PWD/omni-swarm_amd/lib/libomni_hip_$v.so; fi
    timeout 200 python bench.py --no-cpu-baseline --steps 40 --warmup 10 --match-db-rows 8192 --batched-rows 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['stages_ms_per_keyframe']; print('$v', d['value'], s['conv1b+pool'], s['conv2a'], s['conv2b+pool'], s['conv3b+pool'], s['conv4a'], s['convPa|convDa'], r['superpoint_ms_per_keyframe'])"
  done
done
