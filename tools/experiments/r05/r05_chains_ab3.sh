#!/bin/bash
tag=${1:-r05d}
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
run base --steps 60
run chains2_off_optin --steps 60 --chains 2 --defer-da off
run chains2_off_optoff --steps 60 --chains 2 --defer-da off --opt-in-backward off
run chains2_main_optin --steps 60 --chains 2 --defer-da main
run chains2_main_optoff --steps 60 --chains 2 --defer-da main --opt-in-backward off
run chains4_off_optoff --steps 60 --chains 4 --defer-da off --opt-in-backward off
run chains2_normal --steps 60 --chains 2 --defer-da off --opt-in-backward off --chain-priority normal
run base2 --steps 60
run bucket1 --steps 40 --defer-da bucket
run side1 --steps 40 --defer-da side
run off1 --steps 40 --defer-da off
