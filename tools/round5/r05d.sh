#!/bin/bash
# round 5, call 4: the bitmap form of the fused threshold (head: one word per lane; sp_mask_kernel: compaction + masks from aligned 16-byte row loads),
# and where MobileNetVLAD's block kernels spend a tile (OMNI_VLAD_SB_TRACE)
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_sp_post.py tests/test_gpu_superpoint.py -m gpu -q -x > $OUT/r05d_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/r05d_pytest.log)"
timeout 600 python -m pytest tests/test_gpu_bench_shape.py -m gpu -q -x -k "64_images" > $OUT/r05d_pytest2.log 2>&1; echo "pytest2 rc=$? $(tail -1 $OUT/r05d_pytest2.log)"
echo "t=$(( $(date +%s) - T0 ))s"
for P in f16 split; do
  for F in 0 1; do
    echo "== PREC=$P OMNI_SP_FUSED_CAND=$F"
    PREC=$P BATCH=64 NO_VLAD=1 OMNI_SP_PROFILE_MASK=1 OMNI_SP_FUSED_CAND=$F timeout 120 python tools/stage_timing.py 2>&1 | tail -1
  done
done > $OUT/r05d_stage_ab.log 2>&1
cat $OUT/r05d_stage_ab.log
echo "t=$(( $(date +%s) - T0 ))s"
# kernel durations of the post-processing kernels with the fused threshold on
rm -rf $OUT/r05d_tr
PREC=f16 BATCH=64 NO_VLAD=1 OMNI_SP_PROFILE_MASK=1 timeout 200 rocprofv3 --kernel-trace -d $OUT/r05d_tr -o t -- python tools/stage_timing.py > /dev/null 2> $OUT/r05d_tr.err
python tools/rocprof_summary.py $(ls $OUT/r05d_tr/*_results.db $OUT/r05d_tr/*/*_results.db 2>/dev/null | head -1) "(r05d stage_timing f16 batch 64)" | grep -E "sp_|detector|median" | cut -c1-200
rm -rf $OUT/r05d_tr
echo "t=$(( $(date +%s) - T0 ))s"
BATCH=32 PREC=f32 OMNI_VLAD_SB_TRACE=1 timeout 200 python tools/vlad_trace32.py > $OUT/r05d_vlad_trace.log 2>&1
cat $OUT/r05d_vlad_trace.log | cut -c1-250
echo "t=$(( $(date +%s) - T0 ))s"
