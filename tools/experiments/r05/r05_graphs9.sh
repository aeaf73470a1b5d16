#!/bin/bash
# every chain as a linear graph of its own on a stream of its own (timing only, --no-optimizer, dA_m inside the chain)   usage: r05_graphs9.sh <tag>
OUT=gpurun_out/${1:-r05h}; mkdir -p $OUT
N="--no-cpu-baseline --no-traffic --no-optimizer --steps 30"
b() { name=$1; shift; timeout 300 python bench.py $N "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$? $(python -c "
import json,sys
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['adapter_hbm_roofline_frac'], d['chains'])
except Exception as e: print('ERR', e, open('$OUT/$name.err').read()[-400:])
")"; }
b base
b c2off --defer-da off
b c1off --chains 1 --defer-da off
b c1g9 --chains 1 --defer-da off --graphs 9
b c2g9 --chains 2 --defer-da off --graphs 9
b c3g9 --chains 3 --defer-da off --graphs 9
b c4g9 --chains 4 --defer-da off --graphs 9
b b8c4g9 --batch 8 --chains 4 --defer-da off --graphs 9
b b8c2g9 --batch 8 --chains 2 --defer-da off --graphs 9
b b8c2off --batch 8 --chains 2 --defer-da off
GPU_MAX_HW_QUEUES=8 b c4g9_q8 --chains 4 --defer-da off --graphs 9
