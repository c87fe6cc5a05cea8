#!/bin/bash
# builds omni-swarm_amd/lib/libomni_hip_steptrace.so: the library with the Winograd kernel's per-region s_memtime trace compiled in (-DWN_STEP_TRACE); run on the
# GPU box as  OMNI_LIB=$PWD/omni-swarm_amd/lib/libomni_hip_steptrace.so python tools/round6/wino_trace.py
cd "$(dirname "$0")/../../omni-swarm_amd" && make -s lib/libomni_hip.so && mkdir -p build_tr && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize -DWN_STEP_TRACE -c csrc/conv_wino.hip -o build_tr/conv_wino.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libomni_hip_steptrace.so $(ls build/*.o | grep -v "_tv.o\|conv_wino.o") build_tr/conv_wino.o && ls -la lib/
