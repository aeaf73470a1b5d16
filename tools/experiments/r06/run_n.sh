#!/bin/bash
# round 6, GPU call n: BASELINE configs[3] with dropout 0 -- how much of the rank-pad-64 kernels' time is the mask hash (per (kernel, grid) durations, one chain, no graph)
TAG=r6n; REPO=$PWD; mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
for dp in 0.05 0; do
rm -rf /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o r -- python $REPO/bench.py --model 13b --rank 64 --seq 4096 --batch 2 --layers 8 --steps 3 --warmup 2 --no-cpu-baseline --no-traffic --ablate off --defer-da off --graph off --chains 1 --dropout $dp > $REPO/gpurun_out/$TAG/prof_run_$dp.log 2>&1
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
echo "== dropout $dp"; python $REPO/tools/rocpd_summary.py $DB bygrid 0.5 1.0 | grep "moka_" | head -24
done 2>&1 | tee $REPO/gpurun_out/$TAG/dropout.txt
cd $REPO
for dp in 0.05 0; do python bench.py --model 13b --rank 64 --seq 4096 --batch 2 --steps 10 --no-cpu-baseline --no-traffic --ablate off --dropout $dp 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dropout $dp', d['ms_per_step'], d['adapter_hbm_roofline_frac'])"; done | tee -a gpurun_out/$TAG/dropout.txt
