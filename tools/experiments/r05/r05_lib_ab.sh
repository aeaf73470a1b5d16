#!/bin/bash
# same-box A/B of two builds of the library, alternating: moka_amd/libmoka_hip_old.so (MOKA_HIP_LIB) against the default   usage: r05_lib_ab.sh [bench args]
r() { python bench.py --no-cpu-baseline --no-traffic --steps 60 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['adapter_hbm_roofline_frac'])"; }
r "$@" > /dev/null
for i in 1 2 3; do
  echo "old $(MOKA_HIP_LIB=$PWD/moka_amd/libmoka_hip_old.so r "$@")"
  echo "new $(r "$@")"
done
