#!/bin/bash
# PMC passes for roofline.traffic (run on the GPU box from the repo root): gpurun_out/<tag>/{pmc_traffic.json,pmc_fetch_size.md,pmc_write_size.md}
TAG=${1:-pmc}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
LAYERS=4
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcF /tmp/pmcW /tmp/pmc[0-9]*
# --graph off --steps 1 --warmup 1: 2 optimizer steps + the extra bracketed passes; every pass runs the forward once per layer
# --batch 2: the default schedule runs the micro-batch of 4 sequences as two part-batch chains, i.e. launches of 4096 tokens (the launch
# heuristics follow T: the fused up-projection uses 8 column ranges per token block there, 4 at 8192 tokens)
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmcF -o f -- python $REPO/bench.py --layers $LAYERS --steps 1 --warmup 1 --graph off --batch 2 --no-cpu-baseline --no-traffic > $OUT/runF.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmcW -o w -- python $REPO/bench.py --layers $LAYERS --steps 1 --warmup 1 --graph off --batch 2 --no-cpu-baseline --no-traffic > $OUT/runW.log 2>&1
cd $REPO
F=$(find /tmp/pmcF -name "*.db" | head -1); W=$(find /tmp/pmcW -name "*.db" | head -1)
python tools/rocpd_pmc_summary.py $F FETCH_SIZE > $OUT/pmc_fetch_size.md
python tools/rocpd_pmc_summary.py $W WRITE_SIZE > $OUT/pmc_write_size.md
python tools/pmc_traffic.py $F $W $LAYERS 4096 > $OUT/pmc_traffic.json
cat $OUT/pmc_traffic.json | head -12
