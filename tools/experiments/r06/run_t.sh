#!/bin/bash
# round 6, GPU call t: per (kernel, grid) durations, launches alone (no graph, dA_m / dB in the chain), bench layout against the same spans on 128-token
# boundaries -- which kernels pay for token blocks that hold two modalities.  usage: run_t.sh <tag> <bench args>
TAG=$1; shift; REPO=$PWD; mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
for lay in bench aligned; do
rm -rf /tmp/prof_$TAG
MOKA_BENCH_LAYOUT=$lay timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o r -- python $REPO/bench.py --layers 8 --steps 3 --warmup 2 --no-cpu-baseline --no-traffic --ablate off --defer-da off --graph off "$@" > $REPO/gpurun_out/$TAG/prof_run_$lay.log 2>&1
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
echo "== layout $lay"; python $REPO/tools/rocpd_summary.py $DB bygrid 0.5 1.0 | grep "moka_" | head -30
done 2>&1 | tee $REPO/gpurun_out/$TAG/bygrid.txt
