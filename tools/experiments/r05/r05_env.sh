#!/bin/bash
# runtime environment knobs of the hipGraph executor against the default two-chain step (same box)
run() { python bench.py --steps 40 --no-cpu-baseline --no-traffic | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['graph_replay_host_ms'])"; }
run default
for v in 1 4 16 64 256 1024; do DEBUG_HIP_GRAPH_BATCH_SIZE=$v run "DEBUG_HIP_GRAPH_BATCH_SIZE=$v"; done
for v in 0 1; do DEBUG_CLR_GRAPH_PACKET_CAPTURE=$v run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=$v"; done
for v in 0 1; do DEBUG_HIP_DYNAMIC_QUEUES=$v run "DEBUG_HIP_DYNAMIC_QUEUES=$v"; done
for v in 0 1; do AMD_DIRECT_DISPATCH=$v run "AMD_DIRECT_DISPATCH=$v"; done
for v in 0 1; do ROC_ACTIVE_WAIT_TIMEOUT=$v run "ROC_ACTIVE_WAIT_TIMEOUT=$v"; done
HIP_FORCE_DEV_KERNARG=0 run "HIP_FORCE_DEV_KERNARG=0"
DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1 run "DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1"
ROC_SKIP_KERNEL_ARG_COPY=1 run "ROC_SKIP_KERNEL_ARG_COPY=1"
run default
