#!/bin/bash
# same-box A/B of two builds of the library: MOKA_HIP_LIB=old / default, alternating   usage: r05_hash_ab.sh
python tools/probes/dropout_mask_stats.py
r() { python bench.py --no-cpu-baseline --no-traffic --steps 60 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['adapter_hbm_roofline_frac'])"; }
for i in 1 2 3; do
  echo -n "old "; MOKA_HIP_LIB=$PWD/moka_amd/libmoka_hip_old.so r "$@"
  echo -n "new "; r "$@"
done
echo -n "nodrop "; r --dropout 0 "$@"
