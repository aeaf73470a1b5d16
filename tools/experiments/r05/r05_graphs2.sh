#!/bin/bash
# Do TWO hub-shaped graphs on two streams run more than two chains side by side?  (timing only, --no-optimizer)   usage: r05_graphs2.sh <tag>
OUT=gpurun_out/${1:-r05g}; mkdir -p $OUT
N="--no-cpu-baseline --no-traffic --no-optimizer --steps 30"
b() { name=$1; shift; timeout 300 python bench.py $N "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$? $(python -c "
import json,sys
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['adapter_hbm_roofline_frac'], d['chains'])
except Exception as e: print('ERR', e, open('$OUT/$name.err').read()[-400:])
")"; }
b base
b c4g2 --chains 4 --graphs 2
b c4 --chains 4
b c3g2 --chains 3 --graphs 2
b c2g2 --chains 2 --graphs 2
b b8 --batch 8
b b8c4g2 --batch 8 --chains 4 --graphs 2
b b8c8g2 --batch 8 --chains 8 --graphs 2
b b8c3g2 --batch 8 --chains 3 --graphs 2
b b6c3g2 --batch 6 --chains 3 --graphs 2
b b6 --batch 6
