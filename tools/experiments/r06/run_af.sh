#!/bin/bash
# round 6, GPU call af: BASELINE configs[3] (13B, r = 64, 2 x 4096 tokens, one chain): the launch-rule knobs of the diagnostics library, two rounds
mkdir -p gpurun_out/r6af
run() { name=$1; shift; MOKA_HIP_LIB=$PWD/moka_amd/libmoka_hip_diag.so timeout 600 python bench.py --model 13b --rank 64 --seq 4096 --batch 2 --steps 10 --no-cpu-baseline --no-traffic --ablate off "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-34s' % '$name', d['ms_per_step'], d['adapter_hbm_roofline_frac'])"; }
for rep in 1 2; do
run "default"
for kv in expand_bpc=3 expand_bpc=4 expand_bpc=6 dx_group=3 dx_group=5 dx_group=7 gy_ng=2 gy_ng=3 xa_ng=2 xa_ng=4 xa_ng=6 g64_da=1 g64_da=2 wgrad_bpc=2 wgrad_bpc=4 wgrad_ct=2 wgrad_nw=4 cu_div=2 gy_form=1 expand_nq=4; do
MOKA_TUNE=$kv run "$kv"
done; done 2>&1 | tee gpurun_out/r6af/knobs.txt
