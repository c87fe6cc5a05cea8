#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/sparse_da_diag.py > gpurun_out/r04q_diag.log 2>&1
echo "diag rc=$?"; tail -40 gpurun_out/r04q_diag.log
