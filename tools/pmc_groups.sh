#!/bin/bash
# Counter passes of a short bench run, one rocprofv3 --pmc pass per counter group -> gpurun_out/<tag>/summary.md.
# usage: pmc_groups.sh <tag> "<bench args>" "<group 1>" "<group 2>" ...      (run on the GPU box from the repo root)
TAG=$1; BARGS=$2; shift 2
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
rm -f $OUT/errors.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcg_*
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmcg_$i -o p$i -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-traffic --graph off $BARGS > $OUT/run$i.log 2>&1 || echo "pass $i ($grp) failed" >> $OUT/errors.txt
done
python $REPO/tools/rocpd_pmc_multi.py $(find /tmp/pmcg_* -name "*.db") > $OUT/summary.md 2>> $OUT/errors.txt
tail -5 $OUT/errors.txt 2>/dev/null
