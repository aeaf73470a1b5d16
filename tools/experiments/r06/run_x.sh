#!/bin/bash
# round 6, GPU call x: after a kernel change -- the whole GPU suite, smoke(), the PMC traffic summary for the new sources, the three bench lines that moved
mkdir -p gpurun_out/r6x
bash tools/gputests.sh 2>&1 | tail -8
bash tools/pmc_traffic.sh r6x > /dev/null 2>&1; head -c 600 gpurun_out/r6x/pmc_traffic.json; echo
N="--no-cpu-baseline --no-traffic"
timeout 900 python bench.py --model 13b --rank 64 --seq 4096 --batch 2 $N > gpurun_out/r6x/bench_13b_r64_seq4096.json 2>> gpurun_out/r6x/bench.err
timeout 900 python bench.py --rank 32 --steps 40 $N > gpurun_out/r6x/bench_r32.json 2>> gpurun_out/r6x/bench.err
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r6x/bench_driver_command.json 2>> gpurun_out/r6x/bench.err
for f in gpurun_out/r6x/bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['adapter_hbm_roofline_frac'], d['roofline'].get('traffic'))"; done
