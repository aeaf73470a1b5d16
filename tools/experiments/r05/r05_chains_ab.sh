#!/bin/bash
# round 5: --chains as a fair contender (deferred dA per chain, optimizer slices behind every chain's bucket) on ONE box
# usage (gpurun): bash tools/r05_chains_ab.sh <tag>
tag=${1:-r05a}
out=gpurun_out/$tag
mkdir -p $out
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], d['adapter_hbm_roofline_frac'], d.get('forward_only'))
except Exception as e:
    print('ERR', e)
PY
)"; }
run base --steps 60 --probe-forward
run chains2 --steps 60 --chains 2
run chains4 --steps 60 --chains 4
run chains2_side --steps 60 --chains 2 --defer-da side
run chains2_nodefer --steps 60 --chains 2 --defer-da off
run base2 --steps 60
run b1 --steps 100 --batch 1
run b2 --steps 80 --batch 2
run b2_chains2 --steps 80 --batch 2 --chains 2
run b8 --steps 40 --batch 8
run b8_chains2 --steps 40 --batch 8 --chains 2
run b8_chains4 --steps 40 --batch 8 --chains 4
