#!/bin/bash
# Run on the MI355X box (via gpurun): kernel trace + PMC passes of the default bench workload.
#   tools/profile_gpu.sh <tag> [extra bench.py arguments, e.g. --precision split]   -> gpurun_out/<tag>_*  (summaries are copied into profiles/ by hand afterwards)
# PMC passes are separate runs with --kernel-trace only (never combined with sys/hip/hsa traces).
set -u
TAG=${1:-r01x}
shift
EXTRA="$*"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
LEGS="--no-cpu-baseline --f32-steps 0 --python-steps 0 --geometry-steps 0 --big-db-keyframes 0 --parity-steps 0 --c5-rows 0 --long-region-steps 0 $EXTRA"
BENCH="python bench.py --steps 40 --warmup 8 --min-time 0 $LEGS"
BENCH_PMC="python bench.py --steps 8 --warmup 8 --min-time 0 --match-db-rows 100000 $LEGS"
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o ${TAG} -- $BENCH > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_trace.err
python tools/rocprof_summary.py $(ls $OUT/${TAG}_trace/*_results.db $OUT/${TAG}_trace/*/*_results.db 2>/dev/null | head -1) --json $OUT/${TAG}_kernel_times.json "(${TAG}; $BENCH; 1x MI355X)" > $OUT/${TAG}_kernel_stats.md 2>> $OUT/${TAG}_trace.err
i=0
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/${TAG}_pmc$i -o pmc$i -- $BENCH_PMC > $OUT/${TAG}_pmc$i.json 2> $OUT/${TAG}_pmc$i.err
  echo "pmc pass $i ($PMC) rc=$?" >> $OUT/${TAG}_trace.err
done
python tools/pmc_summary.py $OUT/${TAG}_pmc* > $OUT/${TAG}_pmc_summary.md 2>> $OUT/${TAG}_trace.err
python tools/make_traffic.py ${TAG} 64 > $OUT/${TAG}_traffic.json 2>> $OUT/${TAG}_trace.err
# keep the merge under the 64 MiB cap: drop the raw trace DBs, keep CSV counter files only if small
find $OUT/${TAG}_trace -name '*.db' -size +20M -delete
find $OUT -name '*kernel_trace.csv' -size +8M -delete
find $OUT -name '*counter_collection.csv' -size +8M -delete
ls -la $OUT >> $OUT/${TAG}_trace.err
