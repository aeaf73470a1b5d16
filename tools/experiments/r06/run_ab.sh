#!/bin/bash
# round 6, GPU call ab: r = 32, two chains: launch-rule knobs of the diagnostics library around the new defaults (one line per setting, two rounds)
mkdir -p gpurun_out/r6ab
run() { name=$1; shift; MOKA_HIP_LIB=$PWD/moka_amd/libmoka_hip_diag.so timeout 600 python bench.py --rank 32 --steps 20 --no-cpu-baseline --no-traffic --ablate off "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-34s' % '$name', d['ms_per_step'], d['adapter_hbm_roofline_frac'])"; }
for rep in 1 2; do
run "default"
MOKA_TUNE=gy_form=1 run "gy_form=1 (g pass + dB pass)"
MOKA_TUNE=gy_ng=8 run "gy_ng=8"
MOKA_TUNE=gy_ng=16 run "gy_ng=16"
MOKA_TUNE=yx_bpc=2 run "yx_bpc=2"
MOKA_TUNE=yx_bpc=3 run "yx_bpc=3"
MOKA_TUNE=yx_cpb=8 run "yx_cpb=8"
MOKA_TUNE=dx_group=3 run "dx_group=3 (2 per CU)"
MOKA_TUNE=dx_group=5 run "dx_group=5 (4 per CU)"
MOKA_TUNE=expand_bpc=4 run "expand_bpc=4"
MOKA_TUNE=wgrad_bpc=2 run "wgrad_bpc=2"
run "priority normal" --chain-priority normal
run "priority high" --chain-priority high
done 2>&1 | tee gpurun_out/r6ab/knobs.txt
