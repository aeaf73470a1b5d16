#!/bin/bash
out=gpurun_out/r05r64b; mkdir -p $out
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-traffic "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["adapter_hbm_roofline_frac"], "chains", d["chains"], d["defer_dA"], d["defer_dB"], d["entry_point_ms_per_pass"])
except Exception as e:
    print('ERR', e); print(open('$out/$name.err').read()[-600:])
PY
)"; }
R64="--model 13b --rank 64 --seq 4096 --batch 2 --steps 10"
run c1 $R64
run c2_off $R64 --chains 2 --defer-da off
run c2_main $R64 --chains 2 --defer-da main
run c2_layer_nodb $R64 --chains 2 --defer-da layer --defer-db off
run c1_nodb $R64 --defer-db off
run c1_off $R64 --defer-da off
