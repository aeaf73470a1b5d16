#!/bin/bash
tag=${1:-r05g}
out=gpurun_out/$tag
mkdir -p $out
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["adapter_hbm_roofline_frac"], d["optimizer_in_backward"], "host", d["graph_replay_host_ms"])
except Exception as e:
    print('ERR', e); print(open('$out/$name.err').read()[-600:])
PY
)"; }
run old_chain --steps 40 --graph-topology chain
run hub1_layer --steps 40
run hub1_layer_sidefirst --steps 40 --capture-order side-first
run hub1_bucket --steps 40 --defer-da bucket
run hub1_side --steps 40 --defer-da side
run hub1_normal --steps 40 --chain-priority normal
run chains2_layer --steps 40 --chains 2
run chains2_bucket --steps 40 --chains 2 --defer-da bucket
run chains2_off --steps 40 --chains 2 --defer-da off
run chains2_side --steps 40 --chains 2 --defer-da side
run chains4_layer --steps 40 --chains 4
run chains4_off --steps 40 --chains 4 --defer-da off
run old_chain2 --steps 40 --graph-topology chain
run b8_hub1 --steps 30 --batch 8
run b8_chains2 --steps 30 --batch 8 --chains 2
run b8_chains4 --steps 30 --batch 8 --chains 4
run b1_hub1 --steps 60 --batch 1
run b2_hub1 --steps 60 --batch 2
run b2_chains2 --steps 60 --batch 2 --chains 2
