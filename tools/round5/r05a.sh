#!/bin/bash
# round 5, call 1: baselines at the round-4 HEAD for this round's kernel work.
#  (1) MobileNetVLAD alone, 32 images per launch, per-dispatch durations in launch order (the product's default precision)
#  (2) ONE unit in flight (--pipelines 1): per-kernel medians without overlap from other units, fp16 and split
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
bash tools/vlad_seq_trace.sh f32 32 > $OUT/r05a_vlad32_seq.txt 2>&1; echo "vlad seq rc=$?"; tail -3 $OUT/r05a_vlad32_seq.txt
LEGS="--no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --parity-steps 0 --c5-rows 0 --long-region-steps 0 --batched-rows 0"
for P in f16 split; do
  TAG=r05a_${P}_single_unit
  timeout 300 rocprofv3 --kernel-trace -d $OUT/${TAG}_trace -o ${TAG} -- python bench.py --precision $P --pipelines 1 --steps 40 --warmup 8 --min-time 0 $LEGS > $OUT/${TAG}_bench.json 2> $OUT/${TAG}.err
  echo "$P trace rc=$?"
  python tools/rocprof_summary.py $(ls $OUT/${TAG}_trace/*_results.db $OUT/${TAG}_trace/*/*_results.db 2>/dev/null | head -1) --json $OUT/${TAG}_kernel_times.json "(${TAG}: bench.py --precision $P --pipelines 1 --steps 40: one unit of 8 key frames in flight, no overlap between units; 1x MI355X)" > $OUT/${TAG}_kernel_stats.md 2>> $OUT/${TAG}.err
  find $OUT/${TAG}_trace -name '*.db' -size +20M -delete
  head -30 $OUT/${TAG}_kernel_stats.md | cut -c1-220
done
echo "t=$(( $(date +%s) - T0 ))s"
