#!/bin/bash
# round 6, GPU call ai: the runtime's hardware-queue count (GPU_MAX_HW_QUEUES, default 4) against the number of lists a captured step has: hub + 2 chains = 3;
# with 4 chains the lists outnumber the queues -- does a larger pool let four chains run side by side?
mkdir -p gpurun_out/r6ai2
run() { name=$1; shift; timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-traffic --ablate off "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-44s' % '$name', d['ms_per_step'], d['adapter_hbm_roofline_frac'], 'chains', d['chains'])"; }
(
run "default queues, 2 chains"
run "default queues, 4 chains" --chains 4
for q in 8 16; do
GPU_MAX_HW_QUEUES=$q run "GPU_MAX_HW_QUEUES=$q, 2 chains"
GPU_MAX_HW_QUEUES=$q run "GPU_MAX_HW_QUEUES=$q, 4 chains" --chains 4
GPU_MAX_HW_QUEUES=$q run "GPU_MAX_HW_QUEUES=$q, 8 seq, 4 chains" --chains 4 --batch 8
done
run "default queues, 8 seq, 2 chains" --batch 8
run "default queues, 8 seq, 4 chains" --batch 8 --chains 4
GPU_MAX_HW_QUEUES=2 run "GPU_MAX_HW_QUEUES=2, 2 chains"
) 2>&1 | tee gpurun_out/r6ai2/hw_queues.txt
