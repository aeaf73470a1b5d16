#!/bin/bash
# usage: ab_lib.sh "<bench args>" lib1.so lib2.so ...   ("default" = the in-tree build): entry-point split per library
BARGS=$1; shift
mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/ablib.log
for l in "$@"; do
  if [ "$l" = "default" ]; then unset MOKA_HIP_LIB; else export MOKA_HIP_LIB=$PWD/$l; fi
  python bench.py --steps 12 --no-cpu-baseline --no-traffic $BARGS 2>>gpurun_out/ab/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$l\", d[\"ms_per_step\"], d[\"entry_point_ms_per_pass\"])" >> gpurun_out/ab/ablib.log
done
cat gpurun_out/ab/ablib.log
