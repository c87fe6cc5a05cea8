#!/bin/bash
# round 5, call 3: the detector head thresholds its own output (OMNI_SP_FUSED_CAND) and the XCD-aware block ids of the persistent convolution kernels
# (OMNI_CONV_XCD): bit-identity tests, stage times A/B in both precisions, HBM read bytes of the split kernels with and without
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_sp_post.py tests/test_gpu_superpoint.py tests/test_gpu_mask_skip.py -m gpu -q -x > $OUT/r05c_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/r05c_pytest.log)"
timeout 600 python -m pytest tests/test_gpu_bench_shape.py -m gpu -q -x -k "64_images or 32_images" > $OUT/r05c_pytest2.log 2>&1; echo "pytest2 rc=$? $(tail -1 $OUT/r05c_pytest2.log)"
echo "t=$(( $(date +%s) - T0 ))s"
for P in f16 split; do
  for V in "0 0" "1 0" "1 1"; do
    set -- $V
    echo "== PREC=$P OMNI_CONV_XCD=$1 OMNI_SP_FUSED_CAND=$2"
    PREC=$P BATCH=64 NO_VLAD=1 OMNI_SP_PROFILE_MASK=1 OMNI_CONV_XCD=$1 OMNI_SP_FUSED_CAND=$2 timeout 120 python tools/stage_timing.py 2>&1 | tail -1
  done
done > $OUT/r05c_stage_ab.log 2>&1
cat $OUT/r05c_stage_ab.log
echo "t=$(( $(date +%s) - T0 ))s"
for X in 0 1; do
  rm -rf $OUT/r05c_fetch$X
  PREC=split BATCH=64 NO_VLAD=1 OMNI_SP_PROFILE_MASK=1 OMNI_CONV_XCD=$X timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/r05c_fetch$X -o f -- python tools/stage_timing.py > /dev/null 2> $OUT/r05c_fetch$X.err
  echo "== split, OMNI_CONV_XCD=$X: HBM read bytes per launch"; python tools/fetch_by_kernel.py $OUT/r05c_fetch$X conv | grep -v "x  [0-9] " | sort -k9 | tail -24
  rm -rf $OUT/r05c_fetch$X
done > $OUT/r05c_fetch.log 2>&1
cat $OUT/r05c_fetch.log
for X in 0 1; do
  rm -rf $OUT/r05c_fetchh$X
  PREC=f16 BATCH=64 NO_VLAD=1 OMNI_SP_PROFILE_MASK=1 OMNI_CONV_XCD=$X timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/r05c_fetchh$X -o f -- python tools/stage_timing.py > /dev/null 2> $OUT/r05c_fetchh$X.err
  echo "== f16, OMNI_CONV_XCD=$X: HBM read bytes per launch"; python tools/fetch_by_kernel.py $OUT/r05c_fetchh$X conv | sort -k9 | tail -24
  rm -rf $OUT/r05c_fetchh$X
done > $OUT/r05c_fetch_f16.log 2>&1
cat $OUT/r05c_fetch_f16.log
echo "t=$(( $(date +%s) - T0 ))s"
