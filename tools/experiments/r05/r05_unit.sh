#!/bin/bash
tag=${1:-r05u}
out=gpurun_out/$tag
mkdir -p $out
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-traffic "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["adapter_hbm_roofline_frac"], "chains", d["chains"], d["defer_dA"], "host", d["graph_replay_host_ms"], d.get("graph_check"))
except Exception as e:
    print('ERR', e); print(open('$out/$name.err').read()[-600:])
PY
)"; }
run verify_unit --steps 3 --layers 5 --seq 512 --no-optimizer --verify-graph --defer-da unit
run side --steps 100
run unit --steps 100 --defer-da unit
run layer --steps 100 --defer-da layer
run side2 --steps 100
run unit2 --steps 100 --defer-da unit
run b8_side --steps 40 --batch 8
run b8_unit --steps 40 --batch 8 --defer-da unit
run b2_side --steps 100 --batch 2
run b2_unit --steps 100 --batch 2 --defer-da unit
