#!/bin/bash
# round 5, call 8: OMNI_PREC_SPLIT's tails on the fp16 matrix cores with split operands -- the detector head (OMNI_DET16) and convDb + norm at the key
# points' cells (OMNI_SP_SPLIT_DB) -- and the head at two waves per SIMD: parity gates + stage times A/B
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_superpoint.py tests/test_gpu_sp_post.py tests/test_gpu_mask_skip.py -m gpu -q -x > $OUT/r05h_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/r05h_pytest.log)"
timeout 600 python -m pytest tests/test_gpu_bench_shape.py tests/test_gpu_e2e_scene.py tests/test_gpu_e2e_depth.py -m gpu -q -x > $OUT/r05h_pytest2.log 2>&1; echo "pytest2 rc=$? $(tail -1 $OUT/r05h_pytest2.log)"
echo "t=$(( $(date +%s) - T0 ))s"
for V in "0 0" "1 0" "0 1" "1 1"; do
  set -- $V
  echo "== PREC=split OMNI_DET16=$1 OMNI_SP_SPLIT_DB=$2"
  PREC=split BATCH=64 NO_VLAD=1 OMNI_SP_PROFILE_MASK=1 OMNI_DET16=$1 OMNI_SP_SPLIT_DB=$2 timeout 120 python tools/stage_timing.py 2>&1 | tail -1
done > $OUT/r05h_stage_ab.log 2>&1
echo "== PREC=f16" >> $OUT/r05h_stage_ab.log
PREC=f16 BATCH=64 NO_VLAD=1 OMNI_SP_PROFILE_MASK=1 timeout 120 python tools/stage_timing.py 2>&1 | tail -1 >> $OUT/r05h_stage_ab.log
cat $OUT/r05h_stage_ab.log
echo "t=$(( $(date +%s) - T0 ))s"
