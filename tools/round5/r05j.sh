#!/bin/bash
# round 5, call 10: sp_nms_kernel with a thread's candidates in registers; the exchange timing assertion; stage times
set -u
OUT=gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sp_post.py tests/test_gpu_superpoint.py -m gpu -q -x 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_bench_shape.py -m gpu -q -x 2>&1 | tail -2
for P in f16 split; do PREC=$P BATCH=64 NO_VLAD=1 OMNI_SP_PROFILE_MASK=1 timeout 120 python tools/stage_timing.py 2>&1 | tail -1; done
rm -rf $OUT/r05j_tr
PREC=f16 BATCH=64 NO_VLAD=1 OMNI_SP_PROFILE_MASK=1 timeout 200 rocprofv3 --kernel-trace -d $OUT/r05j_tr -o t -- python tools/stage_timing.py > /dev/null 2> $OUT/r05j_tr.err
python tools/rocprof_summary.py $(ls $OUT/r05j_tr/*_results.db $OUT/r05j_tr/*/*_results.db 2>/dev/null | head -1) "(r05j stage_timing f16 batch 64)" | grep -E "sp_|detector|sparse|median" | cut -c1-200
rm -rf $OUT/r05j_tr
