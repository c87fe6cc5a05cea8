#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_shard_rccl.py -m gpu -q -x > $OUT/r04n_pytest_shard.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/r04n_pytest_shard.log)"
bash tools/rehearse_ranks.sh 8 $OUT/r04n_bench_8ranks_one_gpu_stub.json
tail -5 $OUT/r04n_bench_8ranks_one_gpu_stub.err
rocm-smi --showmemuse 2>/dev/null | tail -5
