#!/bin/bash
# Same-box A/B of two builds of the libraries through the default bench loop (boxes differ by +-3 %, so only same-box pairs mean anything):
#   build the baseline into omni-swarm_amd/lib_prev/ (e.g. `git worktree add /tmp/wt <rev> && make -C /tmp/wt/omni-swarm_amd && cp /tmp/wt/omni-swarm_amd/lib/*.so omni-swarm_amd/lib_prev/`),
#   then on the GPU box: bash tools/ab_lib.sh          (lib_prev/ is git-ignored but travels with the snapshot)
B="python bench.py --no-cpu-baseline --big-db-keyframes 0 --f32-steps 0 --geometry-steps 0 --python-steps 0 --min-time 2"
run() { $B 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"; }
echo current; run
mkdir -p /tmp/cur && cp omni-swarm_amd/lib/*.so /tmp/cur/ && cp omni-swarm_amd/lib_prev/*.so omni-swarm_amd/lib/
echo previous; run
cp /tmp/cur/*.so omni-swarm_amd/lib/
