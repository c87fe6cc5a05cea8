#!/bin/bash
# round 5, call 5: MobileNetVLAD's split block kernel -- four-wave stride-2 workgroups (two per CU instead of one) and the persistent grid sized by the
# runtime's occupancy (registers, not only LDS): parity tests + the per-dispatch sequence at 32 images
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_vlad_detector.py -m gpu -q -x -k "vlad or fused_stem or split_fp16 or fp16_operand" > $OUT/r05e_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/r05e_pytest.log)"
timeout 600 python -m pytest tests/test_gpu_bench_shape.py -m gpu -q -x -k "32_images or vlad" > $OUT/r05e_pytest2.log 2>&1; echo "pytest2 rc=$? $(tail -1 $OUT/r05e_pytest2.log)"
echo "t=$(( $(date +%s) - T0 ))s"
bash tools/vlad_seq_trace.sh f32 32 > $OUT/r05e_vlad32_seq.txt 2>&1; cat $OUT/r05e_vlad32_seq.txt | cut -c1-150
BATCH=32 PREC=f32 timeout 200 python tools/vlad_trace32.py 2>&1 | tail -1
echo "t=$(( $(date +%s) - T0 ))s"
