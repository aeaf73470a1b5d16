#!/bin/bash
tag=${1:-r05c3}
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
run c2 --steps 40
run c3 --steps 40 --chains 3
run c3_side --steps 40 --chains 3 --defer-da side
run c3_off --steps 40 --chains 3 --defer-da off
run c2_side --steps 40 --defer-da side
run c2b --steps 40
run b8_c2 --steps 30 --batch 8
run b8_c3 --steps 30 --batch 8 --chains 3
run b6_c2 --steps 30 --batch 6
run b6_c3 --steps 30 --batch 6 --chains 3
run b3_c1 --steps 30 --batch 3 --chains 1
run b3_c2 --steps 30 --batch 3 --chains 2
run b3_c3 --steps 30 --batch 3 --chains 3
