#!/bin/bash
# round 6, GPU call w: BASELINE configs[3] as TWO part-batch chains (one sequence of 4096 tokens each) under the schedule variants, with this round's kernels
mkdir -p gpurun_out/r6w
R="--model 13b --rank 64 --seq 4096 --batch 2 --steps 10 --no-cpu-baseline --no-traffic --ablate off"
run() { name=$1; shift; timeout 600 python bench.py $R "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-34s' % '$name', d['ms_per_step'], d['adapter_hbm_roofline_frac'], 'chains', d['chains'], d['defer_dA'], d['graph_topology'], d['chain_priority'])"; }
(
run "one chain (default)"
run "two chains" --chains 2
run "two chains, priority normal" --chains 2 --chain-priority normal
run "two chains, dA per layer" --chains 2 --defer-da layer
run "two chains, dA in chain" --chains 2 --defer-da off
run "two chains, chain topology" --chains 2 --graph-topology chain
) 2>&1 | tee gpurun_out/r6w/chains2.txt
