#!/bin/bash
# The N > 1 flow of bench.py rehearsed on ONE GPU: every rank on GPU 0, the collectives of csrc/shard.hip through the stand-in for librccl (tests/stub_rccl,
# test infrastructure).  NOT a scaling number: it checks the unique-id broadcast, the W-list merge, all-gather sizes and per-rank memory before a real node does.
#   tools/rehearse_ranks.sh <ranks> <out.json> [extra bench.py arguments]
set -u
N=${1:-8}; OUTF=${2:-gpurun_out/rehearsal.json}; shift; shift
make -s -C tests/stub_rccl
OMNI_BENCH_ONE_GPU=1 OMNI_RCCL_LIB=$PWD/tests/stub_rccl/libstub_rccl.so HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
  --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $N --steps 16 --warmup 8 --min-time 0 --match-db-rows 80000 --batched-rows 0 --no-cpu-baseline "$@" > $OUTF 2> ${OUTF%.json}.err
echo "rehearsal rc=$?"
python - <<PY
import json
try:
    d = json.loads([l for l in open("$OUTF").read().strip().splitlines() if l.startswith("{")][-1])
    print({k: d[k] for k in ("value", "n_gpus", "rccl_ranks", "librccl", "ms_per_step", "loop_candidates_found")}, d["config"]["host_loop"], d["loop_match"])
except Exception as e:
    print("parse failed", e)
PY
