#!/bin/bash
# round 6, GPU call ad: r = 32, two chains: the remaining launch-rule knobs of the diagnostics library (kernel forms and grid shapes), two rounds
mkdir -p gpurun_out/r6ad
run() { name=$1; shift; MOKA_HIP_LIB=$PWD/moka_amd/libmoka_hip_diag.so timeout 600 python bench.py --rank 32 --steps 20 --no-cpu-baseline --no-traffic --ablate off "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-34s' % '$name', d['ms_per_step'], d['adapter_hbm_roofline_frac'])"; }
for rep in 1 2; do
run "default"
for kv in g32_fwd=1 g32_da=1 g32_da=2 g32_dx=3 expand_nq=3 expand_nq=5 expand_nq=6 wgrad_ct=1 wgrad_ct=2 wgrad_nw=4 wgrad_bpc=4 wgrad_bpc=6 xa_ng=2 xa_ng=4 yx_cpb=4 yx_cpb=2 yx_xcd=2 yx_fill=1 gy_ng=4 expand_bpc=2 expand_bpc=8 expand_depth=3; do
MOKA_TUNE=$kv run "$kv"
done; done 2>&1 | tee gpurun_out/r6ad/knobs.txt
