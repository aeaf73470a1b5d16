#!/bin/bash
tag=${1:-r05i}
out=gpurun_out/$tag
mkdir -p $out
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["adapter_hbm_roofline_frac"], "chains", d["chains"], d["graph"], d["optimizer_in_backward"], "host", d["graph_replay_host_ms"], "comm_exposed", d["comm_exposed_ms"])
except Exception as e:
    print('ERR', e); print(open('$out/$name.err').read()[-600:])
PY
)"; }
run default --steps 40
run chains1 --steps 40 --chains 1
run forcecomm --steps 40 --force-comm
run forcecomm_chains1 --steps 40 --force-comm --chains 1
run forcecomm_bf16 --steps 40 --force-comm --comm-bf16
run graph_bwd --steps 40 --graph bwd
run graph_bwd_chains1 --steps 40 --graph bwd --chains 1
run b1 --steps 60 --batch 1
run b2 --steps 60 --batch 2
run side --steps 40 --defer-da side
run forcecomm_high --steps 40 --force-comm --chain-priority high
run forcecomm_side --steps 40 --force-comm --defer-da side
run r32 --steps 20 --rank 32
run r32_chains1 --steps 20 --rank 32 --chains 1
run 70b --steps 6 --model 70b
run 70b_chains1 --steps 6 --model 70b --chains 1
run 13b --steps 10 --model 13b --rank 64 --seq 4096 --batch 2
run 13b_chains1 --steps 10 --model 13b --rank 64 --seq 4096 --batch 2 --chains 1
