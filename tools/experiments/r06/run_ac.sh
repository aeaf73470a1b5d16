#!/bin/bash
# round 6, GPU call ac: workgroups per CU of the fused up-projection (yx_bpc) at r = 32 and r = 16 (two chains of 4096 tokens), alone and with dx_group = 3
mkdir -p gpurun_out/r6ac
run() { name=$1; shift; MOKA_HIP_LIB=$PWD/moka_amd/libmoka_hip_diag.so timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-traffic --ablate off "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-34s' % '$name', d['ms_per_step'], d['adapter_hbm_roofline_frac'])"; }
for rep in 1 2; do
for r in 32 16; do
run "r$r default" --rank $r
for b in 2 3 4 6; do MOKA_TUNE=yx_bpc=$b run "r$r yx_bpc=$b" --rank $r; done
MOKA_TUNE=yx_bpc=3,dx_group=3 run "r$r yx_bpc=3 dx_group=3" --rank $r
done; done 2>&1 | tee gpurun_out/r6ac/yx_bpc.txt
