#!/bin/bash
# same-box sweep of MOKA_TUNE strings (diagnostics build) on the default (2-chain) schedule
export MOKA_HIP_LIB=$PWD/moka_amd/libmoka_hip_diag.so
tag=${1:-r05t}; shift
out=gpurun_out/$tag; mkdir -p $out
for t in "$@"; do
  if [ "$t" = "none" ]; then tt=""; else tt=$t; fi
  MOKA_TUNE=$tt timeout 300 python bench.py --steps 40 --no-cpu-baseline --no-traffic 2>>$out/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$t\", d[\"value\"], d[\"ms_per_step\"], d[\"entry_point_ms_per_pass\"])" | tee -a $out/tune.log
done
