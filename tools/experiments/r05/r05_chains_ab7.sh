#!/bin/bash
tag=${1:-r05h}
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
run chains2_layer --steps 40 --chains 2
run chains2_layer_normal --steps 40 --chains 2 --chain-priority normal
run chains2_side --steps 40 --chains 2 --defer-da side
run chains2_side_normal --steps 40 --chains 2 --defer-da side --chain-priority normal
run chains2_layer_sidefirst --steps 40 --chains 2 --capture-order side-first
run chains2_nodrop --steps 40 --chains 2 --dropout 0
run chains2_vt --steps 40 --chains 2 --variant vt
run chains4_layer_normal --steps 40 --chains 4 --chain-priority normal
export GPU_MAX_HW_QUEUES=8
echo "== 8 queues"
run q8_old_chain --steps 40 --graph-topology chain
run q8_chains2_layer --steps 40 --chains 2
run q8_chains2_side --steps 40 --chains 2 --defer-da side
run q8_chains4_layer --steps 40 --chains 4
run q8_chains4_layer_normal --steps 40 --chains 4 --chain-priority normal
run q8_b8_chains4 --steps 30 --batch 8 --chains 4
run q8_b8_chains2 --steps 30 --batch 8 --chains 2
