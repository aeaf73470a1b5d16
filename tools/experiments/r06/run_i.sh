#!/bin/bash
# NOT MERGED (the "xs_tpb" knob it drives was reverted: see the result below).
# round 6, GPU call i: the down-projection's tiles-per-workgroup under the two-chain schedule (diagnostics build, "xs_tpb"), same-box alternating pairs
export MOKA_HIP_LIB=$PWD/moka_amd/libmoka_hip_diag.so
run() { MOKA_TUNE=$1 python bench.py --steps 40 --no-cpu-baseline --no-traffic --ablate ${2:-off} 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); a = d.get('ablation', {}).get('families', {})
print('$1', d['ms_per_step'], d['adapter_hbm_roofline_frac'], d['entry_point_ms_per_pass']['moka_down_fwd'], {k: v['marginal_ms'] for k, v in a.items()})"; }
for rep in 1 2 3; do
  run ""
  run "xs_tpb=1"
  run "xs_tpb=16"
  run "xs_tpb=32"
done
run "" down_fwd
run "xs_tpb=16" down_fwd

# result (one box, three alternating rounds, ms per step / moka_down_fwd ms per pass):
#   default (8 tiles per workgroup, >= one workgroup per CU)   29.44 / 29.37 / 29.55    5.79 / 5.72 / 5.75
#   16 tiles where company > 1 (half the workgroups)           29.85 / 29.84 / 30.00    7.47 / 7.54 / 7.50
#   16 tiles always                                            29.87 / 29.90 / 30.07    7.44 / 7.44 / 7.52
#   32 tiles                                                   31.60 / 31.70 / 31.85   11.62
# in-schedule marginal of the family: 3.40 -> 3.95 ms.  Fewer, longer workgroups lose also beside a second chain: the rule stays.
