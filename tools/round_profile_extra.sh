#!/bin/bash
# The other BASELINE configurations on the GPU box (rank 64 / 13B / seq 4096, 70B widths, rank 32) -> gpurun_out/<tag>/; run after round_profile.sh.
TAG=${1:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py --model 13b --rank 64 --seq 4096 --batch 2 --steps 10 --no-cpu-baseline --no-traffic > $OUT/bench_13b_r64_seq4096.json 2>> $OUT/bench.err
python bench.py --model 70b --steps 6 --no-cpu-baseline --no-traffic > $OUT/bench_70b.json 2>> $OUT/bench.err
python bench.py --rank 32 --steps 6 --no-cpu-baseline --no-traffic > $OUT/bench_r32.json 2>> $OUT/bench.err
for f in bench_13b_r64_seq4096 bench_70b bench_r32; do python -c "import json; d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['adapter_hbm_roofline_frac'])"; done
