#!/bin/bash
tag=${1:-r05s}
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
for mb in 32 64 128 256 512 1024; do
run stag$mb --steps 100 --chain-stagger $mb
done
run c2b --steps 100
run side_stag128 --steps 100 --defer-da side --chain-stagger 128
run side --steps 100 --defer-da side
