#!/bin/bash
# round 6, GPU call j: BASELINE configs[3] (13B, r = 64, seq 4096) under schedule variants round 5 did not try: ONE chain in the hub shape with the dA_m / dB
# launches per unit (dB of a unit then starts right behind that unit's pass over gy: does its second read of gy come out of the Infinity Cache?)
R="--model 13b --rank 64 --seq 4096 --batch 2 --steps 10 --no-cpu-baseline --no-traffic --ablate off"
run() { name=$1; shift; python bench.py $R "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-34s' % '$name', d['ms_per_step'], d['adapter_hbm_roofline_frac'], 'chains', d['chains'], d['defer_dA'], d['defer_dB'], d['graph_topology'], d['chain_priority'])"; }
for rep in 1 2; do
run default
run "hub unit" --chains 1 --graph-topology hub --defer-da unit
run "hub layer" --chains 1 --graph-topology hub --defer-da layer
run "hub side" --chains 1 --graph-topology hub --defer-da side
run "chain side" --chains 1 --defer-da side
run "priority normal" --chain-priority normal
run "hub unit dB in chain" --chains 1 --graph-topology hub --defer-da unit --defer-db off
done

# result (one box, two rounds; ms per step):  default (chain topology, dA_m / dB one launch each per layer on the side stream, chain at HIGH priority) 78.05 / 78.00;
#   chain at NORMAL priority 77.37 / 77.52 (adopted: --chain-priority auto = normal above rank 32);  per-unit launches on the side stream 78.66 / 78.34;
#   ONE chain in the hub shape: per unit 98.5, per layer 103.5 / 102.1, at the layer's end 103.5 / 101.9, dB in the chain 109.6 / 110.1 -- the hub list and the one
#   chain share a hardware queue in this process (the same 104 ms the in-schedule ablation of this configuration saw on four of its captures): not a way to start a
#   unit's dB behind its own pass over gy.
