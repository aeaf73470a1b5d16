#!/bin/bash
# usage: ab_tune.sh "<bench args>" tune1 tune2 ...   ("none" = defaults): entry-point split of bench.py per MOKA_TUNE string
# moka_tune lives in the diagnostics build only (the product library keeps no mutable state)
export MOKA_HIP_LIB=${MOKA_HIP_LIB:-$PWD/moka_amd/libmoka_hip_diag.so}
BARGS=$1; shift
mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/abtune.log
for t in "$@"; do
  if [ "$t" = "none" ]; then tt=""; else tt=$t; fi
  MOKA_TUNE=$tt python bench.py --steps 10 --no-cpu-baseline --no-traffic $BARGS 2>>gpurun_out/ab/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$t\", d[\"value\"], d[\"ms_per_step\"], d[\"entry_point_ms_per_pass\"])" >> gpurun_out/ab/abtune.log
done
cat gpurun_out/ab/abtune.log
