#!/bin/bash
# round 4: PINHOLE_DEPTH end to end + the tests around the pipeline / cam unit
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_e2e_depth.py tests/test_gpu_e2e_scene.py tests/test_gpu_bench_shape.py -x -q -m gpu > gpurun_out/r04o_pytest.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r04o_pytest.log
