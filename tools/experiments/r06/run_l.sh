#!/bin/bash
# round 6, GPU call l: kernel trace of BASELINE configs[3] (13B, r = 64, seq 4096, 2 sequences) with EVERYTHING in the one chain (--defer-da off --graph off):
# per (kernel, grid) durations of launches that run alone -> which projection of which family is furthest from its stream ceiling.
TAG=r6l; REPO=$PWD; mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o r -- python $REPO/bench.py --model 13b --rank 64 --seq 4096 --batch 2 --layers 8 --steps 3 --warmup 2 --no-cpu-baseline --no-traffic --ablate off --defer-da off --graph off --chains 1 > $REPO/gpurun_out/$TAG/prof_run.log 2>&1
cd $REPO
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
python tools/rocpd_summary.py $DB bygrid 0.5 1.0 > gpurun_out/$TAG/kernel_trace_bygrid.md 2>&1
tail -1 gpurun_out/$TAG/prof_run.log | cut -c1-200
cat gpurun_out/$TAG/kernel_trace_bygrid.md
