#!/bin/bash
# roctx ranges (OMNI_ROCTX=1) around the stages of the hot path and the host loop: rocprofv3 --marker-trace + --kernel-trace of a short bench run; enable_perf through the C++ adapters
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
python -m pytest tests/test_cpp_host.py -q -x -m gpu 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
OMNI_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --stats -d $R/gpurun_out/r06d_roctx -o roctx --output-format csv -- python $R/bench.py --steps 64 --warmup 16 --no-cpu-baseline --big-db-keyframes 0 --f32-steps 0 --c5-rows 0 --parity-steps 0 --geometry-steps 0 --python-steps 0 --match-db-rows 8192 --batched-rows 0 --long-region-steps 0 --min-time 0.3 > $R/gpurun_out/r06d_bench_under_rocprof.json 2> $R/gpurun_out/r06d_err.log
cd $R; find gpurun_out/r06d_roctx -name "*.csv" | head; for f in $(find gpurun_out/r06d_roctx -name "*marker*stats*.csv"); do echo "== $f"; head -40 $f; done
