#!/bin/bash
# round 6, GPU call o: the dispatch SEQUENCE of one step of BASELINE configs[3] (one chain, no graph): moka_yt_kernel<64> on the 13824-wide projections averages
# 148 us with a minimum of 108 -- which launches are the slow ones?
TAG=r6o; REPO=$PWD; mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o r -- python $REPO/bench.py --model 13b --rank 64 --seq 4096 --batch 2 --layers 4 --steps 3 --warmup 2 --no-cpu-baseline --no-traffic --ablate off --defer-da off --graph off --chains 1 "$@" > $REPO/gpurun_out/$TAG/prof_run.log 2>&1
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
python $REPO/tools/rocpd_summary.py $DB seq 0.70 260 | grep -v "at::native\|rocprim\|rocclr" > $REPO/gpurun_out/$TAG/sequence.md
cat $REPO/gpurun_out/$TAG/sequence.md
