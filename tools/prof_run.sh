#!/bin/bash
# rocprofv3 kernel trace of bench.py on the GPU box -> gpurun_out/<tag>/{kernel_trace.md,timeline.md}.  usage: prof_run.sh <tag> [bench args]
TAG=$1; shift
REPO=$PWD
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o r -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --ablate off "$@" > $REPO/gpurun_out/$TAG/prof_run.log 2>&1
cd $REPO
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
python tools/rocpd_summary.py $DB 40 > gpurun_out/$TAG/kernel_trace.md 2>&1
python tools/rocpd_timeline.py $DB 0.4 > gpurun_out/$TAG/timeline.md 2>&1
# graph replays only (the bracketed live passes behind the timed region run the in-chain schedule): dispatches 15 % .. 50 %
python tools/rocpd_timeline.py $DB 0.85 0.5 > gpurun_out/$TAG/timeline_graph.md 2>&1
tail -2 gpurun_out/$TAG/prof_run.log | cut -c1-300
cat gpurun_out/$TAG/timeline.md
