#!/bin/bash
# round 4: the whole GPU suite + smoke after the config-table refactor
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04s_pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/r04s_pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04s_smoke.log 2>&1
echo "smoke rc=$?"; tail -3 gpurun_out/r04s_smoke.log
