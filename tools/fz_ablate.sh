#!/bin/bash
# builds omni-swarm_amd/lib_abl/libomni_hip_abl<N>.so for the timing ablations of the FUSE1A build (csrc/conv_split.hip, FZ_ABL; WRONG results by design):
#   tools/fz_ablate.sh 1 2 4 8 16    then on the GPU box: OMNI_LIB=omni-swarm_amd/lib_abl/libomni_hip_abl4.so python bench.py --precision split ...
set -e
cd "$(dirname "$0")/../omni-swarm_amd"
mkdir -p lib_abl build_abl
for N in "$@"; do
  DEF="-DFZ_ABL=$N"; [ "$N" = trace ] && DEF="-DSPL_STEP_TRACE"        # trace: s_memtime stamps inside the stream (OMNI_SPLIT_TRACE=1 prints them)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $DEF -c csrc/conv_split.hip -o build_abl/conv_split_$N.o
  OBJS=$(ls build/*.o | grep -v conv_split.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib_abl/libomni_hip_abl$N.so $OBJS build_abl/conv_split_$N.o
done
