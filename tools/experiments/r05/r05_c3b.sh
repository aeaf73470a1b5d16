#!/bin/bash
tag=${1:-r05c3b}
out=gpurun_out/$tag
mkdir -p $out
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-traffic "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["adapter_hbm_roofline_frac"], "chains", d["chains"], d["graph"], "host", d["graph_replay_host_ms"])
except Exception as e:
    print('ERR', e); print(open('$out/$name.err').read()[-600:])
PY
)"; }
run c2 --steps 100
run c3_off_optoff --steps 100 --chains 3 --defer-da off --opt-in-backward off
run c2_off_optoff --steps 100 --chains 2 --defer-da off --opt-in-backward off
run c4_off_optoff --steps 100 --chains 4 --defer-da off --opt-in-backward off
run c2_side --steps 100 --defer-da side
run c2b --steps 100
run b8_c3_off_optoff --steps 40 --batch 8 --chains 3 --defer-da off --opt-in-backward off
run b8_c4_off_optoff --steps 40 --batch 8 --chains 4 --defer-da off --opt-in-backward off
run b8_c2 --steps 40 --batch 8
