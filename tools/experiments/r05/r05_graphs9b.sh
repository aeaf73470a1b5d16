#!/bin/bash
# r05_graphs9.sh with a priority per chain (do the priority levels have queue pools of their own?)   usage: r05_graphs9b.sh <tag>
OUT=gpurun_out/${1:-r05i}; mkdir -p $OUT
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
N="--no-cpu-baseline --no-traffic --no-optimizer --steps 30 --defer-da off --graphs 9"
b() { name=$1; shift; timeout 300 python bench.py $N "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$? $(python -c "
import json,sys
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['adapter_hbm_roofline_frac'], d['chains'])
except Exception as e: print('ERR', e, open('$OUT/$name.err').read()[-400:])
")"; }
b c2 --chains 2
b c4norm --chains 4 --chain-priority normal
b c3norm --chains 3 --chain-priority normal
MOKA_BENCH_PRI_MIX=-1,-1,0,0 b c4mix --chains 4
MOKA_BENCH_PRI_MIX=-1,0,-1,0 b c4mix2 --chains 4
MOKA_BENCH_PRI_MIX=-1,-1,0 b c3mix --chains 3
MOKA_BENCH_PRI_MIX=-1,0,1 b c3mix3 --chains 3
MOKA_BENCH_PRI_MIX=-1,0,1,1 b c4mix3 --chains 4
MOKA_BENCH_PRI_MIX=-1,-1,0,0 b b8c4mix --chains 4 --batch 8
MOKA_BENCH_PRI_MIX=-1,-1,0,0 GPU_MAX_HW_QUEUES=8 b c4mix_q8 --chains 4
