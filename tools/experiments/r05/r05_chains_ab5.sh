#!/bin/bash
tag=${1:-r05f}
out=gpurun_out/$tag
mkdir -p $out
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["adapter_hbm_roofline_frac"], d["optimizer_in_backward"], "host", d["graph_replay_host_ms"])
except Exception as e:
    print('ERR', e)
PY
)"; }
run base_sidefirst --steps 40 --capture-order side-first
run base --steps 40
run bucket1 --steps 40 --defer-da bucket
run side1 --steps 40 --defer-da side
run chains2_off --steps 40 --chains 2 --defer-da off
run chains2_layer --steps 40 --chains 2 --defer-da layer
run chains2_bucket --steps 40 --chains 2 --defer-da bucket
run chains2_off_optoff --steps 40 --chains 2 --defer-da off --opt-in-backward off
run chains4_off --steps 40 --chains 4 --defer-da off
run chains4_layer --steps 40 --chains 4 --defer-da layer
export GPU_MAX_HW_QUEUES=8
echo "== 8 queues"
run q8_base --steps 40
run q8_chains2_off --steps 40 --chains 2 --defer-da off
run q8_chains2_layer --steps 40 --chains 2 --defer-da layer
run q8_chains2_bucket --steps 40 --chains 2 --defer-da bucket
run q8_chains4_layer --steps 40 --chains 4 --defer-da layer
