#!/bin/bash
# usage: ab_units.sh "<bench args>" <entry point> tune1 tune2 ...   ("none" = defaults): per-unit avg_ms of one entry point per MOKA_TUNE string
export MOKA_HIP_LIB=${MOKA_HIP_LIB:-$PWD/moka_amd/libmoka_hip_diag.so}
BARGS=$1; EP=$2; shift; shift
for t in "$@"; do
  if [ "$t" = "none" ]; then tt=""; else tt=$t; fi
  MOKA_TUNE=$tt python bench.py --steps 10 --no-cpu-baseline --no-traffic $BARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$t', d['ms_per_step'], {n.split('[')[1].split(':')[0]: round(v['avg_ms']*1e3,1) for n,v in k.items() if n.startswith('$EP[')})"
done
