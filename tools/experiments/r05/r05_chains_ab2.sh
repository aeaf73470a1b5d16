#!/bin/bash
tag=${1:-r05c}
out=gpurun_out/$tag
mkdir -p $out
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], d['adapter_hbm_roofline_frac'])
except Exception as e:
    print('ERR', e)
PY
)"; }
run base --steps 60
run bucket1 --steps 60 --defer-da bucket
run chains2_bucket --steps 60 --chains 2 --defer-da bucket
run chains2_nodefer --steps 60 --chains 2 --defer-da off
run chains2_bucket_noopt --steps 60 --chains 2 --defer-da bucket --opt-in-backward off
run chains4_bucket --steps 60 --chains 4 --defer-da bucket
run base2 --steps 60
run b8_chains2_bucket --steps 40 --batch 8 --chains 2 --defer-da bucket
run b8_chains4_bucket --steps 40 --batch 8 --chains 4 --defer-da bucket
run b8_chains2_nodefer --steps 40 --batch 8 --chains 2 --defer-da off
run b2_chains2_bucket --steps 80 --batch 2 --chains 2 --defer-da bucket
