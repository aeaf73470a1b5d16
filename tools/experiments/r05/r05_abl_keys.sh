#!/bin/bash
# same-box upper bounds for taking rank-space backward launches off the chain (TIMING ONLY: the ablation builds give wrong key rows):
#   abl  = moka_cross_bwd without its key-row launch (moka_cross_bwd_keys_kernel)   abl2 = without either launch
# built from sed-edited copies of the kernel source; usage: r05_abl_keys.sh [bench args]
mkdir -p gpurun_out/abl
r() { python bench.py --no-cpu-baseline --no-traffic --steps 60 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['adapter_hbm_roofline_frac'], d['entry_point_ms_per_pass'])"; }
r "$@" > /dev/null
for i in 1 2 3; do
  echo "base $(r "$@")"
  echo "abl  $(MOKA_HIP_LIB=$PWD/moka_amd/libmoka_hip_abl.so r "$@")"
  echo "abl2 $(MOKA_HIP_LIB=$PWD/moka_amd/libmoka_hip_abl2.so r "$@")"
done 2>&1 | tee gpurun_out/abl/keys_ab.txt
