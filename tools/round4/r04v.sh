#!/bin/bash
mkdir -p gpurun_out
for Q in 4 8; do for P in 0 1; do
  GPU_MAX_HW_QUEUES=$Q OMNI_HW_QUEUES=0 PROBE_PRIO=$P timeout 300 python tools/queue_probe.py 2>/dev/null | tail -2
done; done > gpurun_out/r04v_probe.log
GPU_MAX_HW_QUEUES=8 OMNI_HW_QUEUES=0 PROBE_PRIO=1 PROBE_EXTRA_STREAMS=0 timeout 300 python tools/queue_probe.py 2>/dev/null | tail -1 >> gpurun_out/r04v_probe.log
cat gpurun_out/r04v_probe.log
