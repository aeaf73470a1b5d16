#!/bin/bash
REPO=$PWD; OUT=$REPO/gpurun_out/pmc64; mkdir -p $OUT
ARGS="--model 13b --rank 64 --seq 4096 --batch 2 --layers 2 --steps 1 --warmup 1 --graph off --no-cpu-baseline --no-traffic --defer-da off"
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmcF /tmp/pmcW
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmcF -o f -- python $REPO/bench.py $ARGS > $OUT/runF.log 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmcW -o w -- python $REPO/bench.py $ARGS > $OUT/runW.log 2>&1
cd $REPO
F=$(find /tmp/pmcF -name "*.db" | head -1); W=$(find /tmp/pmcW -name "*.db" | head -1)
python tools/rocpd_pmc_summary.py $F FETCH_SIZE > $OUT/pmc_fetch_size.md
python tools/rocpd_pmc_summary.py $W WRITE_SIZE > $OUT/pmc_write_size.md
head -40 $OUT/pmc_fetch_size.md | cut -c1-200
