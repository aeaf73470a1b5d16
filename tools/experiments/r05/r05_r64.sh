#!/bin/bash
tag=${1:-r05r64}
out=gpurun_out/$tag
mkdir -p $out
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-traffic "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["adapter_hbm_roofline_frac"], "chains", d["chains"], d["defer_dA"], d["graph_topology"], d["chain_priority"], "host", d["graph_replay_host_ms"])
except Exception as e:
    print('ERR', e); print(open('$out/$name.err').read()[-600:])
PY
)"; }
R64="--model 13b --rank 64 --seq 4096 --batch 2 --steps 12"
run r64_default $R64
run r64_sidefirst $R64 --capture-order side-first
run r64_hub_high $R64 --graph-topology hub
run r64_hub_normal $R64 --graph-topology hub --chain-priority normal
run r64_hub_unit $R64 --graph-topology hub --chain-priority normal --defer-da unit
run r64_c2_unit $R64 --chains 2 --defer-da unit
run r64_c2_layer $R64 --chains 2 --defer-da layer
run r64_c2_layer_normal $R64 --chains 2 --defer-da layer --chain-priority normal
run r64_default2 $R64
R32="--rank 32 --steps 20"
run r32_c1 $R32 --chains 1
run r32_c2 $R32
run r32_c2_layer $R32 --defer-da layer
B70="--model 70b --steps 6"
run 70b_c1 $B70
run 70b_hub_normal $B70 --graph-topology hub --chain-priority normal
