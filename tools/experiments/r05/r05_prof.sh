#!/bin/bash
# rocprofv3 kernel trace of the default bench command -> kernel_trace.md, timeline_graph.md, overlap.md   usage: r05_prof.sh <tag> [bench args]
TAG=$1; shift
REPO=$PWD
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o r -- python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $REPO/gpurun_out/$TAG/prof_run.log 2>&1
cd $REPO
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
python tools/rocpd_summary.py $DB 40 > gpurun_out/$TAG/kernel_trace.md 2>&1
python tools/rocpd_timeline.py $DB 0.85 0.5 > gpurun_out/$TAG/timeline_graph.md 2>&1
python tools/rocpd_overlap.py $DB 0.2 0.5 > gpurun_out/$TAG/overlap.md 2>&1
tail -1 gpurun_out/$TAG/prof_run.log | cut -c1-200
cat gpurun_out/$TAG/overlap.md
