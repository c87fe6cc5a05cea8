#!/bin/bash
# MobileNetVLAD under the fisheye mask: the constant region left out of the tile walk of the stem and the first blocks (OMNI_VLAD_MASK_SKIP):
# bit identity + the other MobileNetVLAD tests, the per-dispatch sequence at 32 images with and without it
mkdir -p gpurun_out
{
timeout 1200 python -m pytest tests/test_gpu_vlad_detector.py -q -x -m gpu -k "vlad or masked or split_fp16 or fused_stem or fp16_operand" 2>&1 | tail -4
for m in 0 1; do
  echo "== OMNI_VLAD_MASK_SKIP=$m"
  OMNI_VLAD_MASK_SKIP=$m bash tools/vlad_seq_trace.sh f32 32 2>&1 | grep -v "^columns" | cut -c1-150
done
for m in 0 1 0 1; do OMNI_VLAD_MASK_SKIP=$m python tools/vlad_trace32.py 2>/dev/null | sed "s/^/mask_skip=$m /"; done
} 2>&1 | tee gpurun_out/r06h_vlad_mask_skip.log
