#!/bin/bash
# round 6, GPU call r: A/B of two libraries on the bench (libmoka_hip_base.so = before this series of changes): BASELINE configs[3], r = 32, the default
mkdir -p gpurun_out/r6r
run() { name=$1; lib=$2; shift 2; MOKA_HIP_LIB=$lib timeout 600 python bench.py --steps 10 --no-cpu-baseline --no-traffic --ablate off "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s' % '$name', d['ms_per_step'], d['adapter_hbm_roofline_frac'])"; }
NEW=$PWD/moka_amd/libmoka_hip.so; OLD=$PWD/moka_amd/libmoka_hip_base.so
for rep in 1 2; do
for v in OLD NEW; do
run "13b r64 s4096 b2 $v" ${!v} --model 13b --rank 64 --seq 4096 --batch 2
run "7b r32 $v" ${!v} --rank 32
[ "$FULL" = 1 ] && run "default $v" ${!v} --steps 40
done; done 2>&1 | tee gpurun_out/r6r/ab.txt
