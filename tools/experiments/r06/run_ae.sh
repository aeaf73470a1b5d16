#!/bin/bash
# round 6, GPU call ae: r = 32, two chains: schedule variants around the default (product library), two rounds
mkdir -p gpurun_out/r6ae
run() { name=$1; shift; timeout 600 python bench.py --rank 32 --steps 20 --no-cpu-baseline --no-traffic --ablate off "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-34s' % '$name', d['ms_per_step'], d['adapter_hbm_roofline_frac'], d['defer_dA'], d['chain_priority'])"; }
for rep in 1 2; do
run "default"
run "dA per layer" --defer-da layer
run "dA at layer end (side)" --defer-da side
run "dA in chain" --defer-da off
run "fuse-fwd off" --fuse-fwd off
run "company hint off" --company-hint off
run "stagger 64 MB" --chain-stagger 64
run "stagger 256 MB" --chain-stagger 256
run "one chain" --chains 1
run "shadows main" --shadows main
done 2>&1 | tee gpurun_out/r6ae/schedule.txt
