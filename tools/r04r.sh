#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_e2e_depth.py tests/test_gpu_bench_shape.py -x -q -m gpu -k "pinhole_depth or cpp_host_loop" > gpurun_out/r04r_pytest.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r04r_pytest.log
