#!/bin/bash
# same-box A/B of two builds of the library through bench.py: MOKA_HIP_LIB=<base> vs the in-tree build.  usage: ab_bench.sh <base.so> [bench args]
BASE=$1; shift
mkdir -p gpurun_out
for rep in 1 2; do
for lib in base new; do
  if [ $lib = base ]; then export MOKA_HIP_LIB=$BASE; else unset MOKA_HIP_LIB; fi
  python bench.py --no-cpu-baseline --steps 6 "$@" > gpurun_out/ab_${lib}_$rep.json
  python - <<PY
import json
d = json.load(open("gpurun_out/ab_${lib}_$rep.json"))
print("$lib", $rep, d["value"], d["ms_per_step"], d["entry_point_ms_per_pass"])
PY
done; done
