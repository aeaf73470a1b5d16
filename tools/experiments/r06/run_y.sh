#!/bin/bash
# round 6, GPU call y: r = 32, two chains of 4096 tokens: the chunk-walk down-projection writes 16 split-K slices per 4096 columns there (three workgroups per CU
# of 32 token blocks -> one 256-column chunk per workgroup); fewer, longer walks (xa_ng = workgroups per CU) halve the slices the fused up-projection re-reads
mkdir -p gpurun_out/r6y
run() { name=$1; shift; MOKA_HIP_LIB=$PWD/moka_amd/libmoka_hip_diag.so timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-traffic --ablate off "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-34s' % '$name', d['ms_per_step'], d['adapter_hbm_roofline_frac'])"; }
for rep in 1 2; do
for ng in 0 1 2; do MOKA_TUNE=xa_ng=$ng run "r32 xa_ng=$ng" --rank 32; done
for ng in 0 1 2; do MOKA_TUNE=xa_ng=$ng run "13b r64 2 chains xa_ng=$ng" --model 13b --rank 64 --seq 4096 --batch 2 --chains 2 --defer-da layer --steps 10; done
done 2>&1 | tee gpurun_out/r6y/xa_ng.txt
