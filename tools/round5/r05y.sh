#!/bin/bash
# round 5 evidence, part 1: rocprofv3 kernel trace + four PMC passes at HEAD, fp16 and split (tools/profile_gpu.sh), and the single-unit (no overlap
# between units) kernel traces of both precisions
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
bash tools/profile_gpu.sh r05yf > $OUT/r05yf_run.log 2>&1; echo "f16 profile rc=$?"
bash tools/profile_gpu.sh r05ys --precision split > $OUT/r05ys_run.log 2>&1; echo "split profile rc=$?"
echo "t=$(( $(date +%s) - T0 ))s"
LEGS="--no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --parity-steps 0 --c5-rows 0 --long-region-steps 0 --batched-rows 0"
for P in f16 split; do
  TAG=r05y_${P}_single_unit
  timeout 300 rocprofv3 --kernel-trace -d $OUT/${TAG}_trace -o ${TAG} -- python bench.py --precision $P --pipelines 1 --steps 40 --warmup 8 --min-time 0 $LEGS > $OUT/${TAG}_bench.json 2> $OUT/${TAG}.err
  python tools/rocprof_summary.py $(ls $OUT/${TAG}_trace/*_results.db $OUT/${TAG}_trace/*/*_results.db 2>/dev/null | head -1) --json $OUT/${TAG}_kernel_times_su.json "(${TAG}: bench.py --precision $P --pipelines 1 --steps 40: one unit of 8 key frames in flight, no overlap between units; 1x MI355X)" > $OUT/${TAG}_kernel_stats.md 2>> $OUT/${TAG}.err
  find $OUT/${TAG}_trace -name '*.db' -size +20M -delete
done
head -20 $OUT/r05yf_kernel_stats.md | cut -c1-200
cat $OUT/r05yf_traffic.json | head -40
cat $OUT/r05ys_traffic.json | head -60
echo "t=$(( $(date +%s) - T0 ))s"
