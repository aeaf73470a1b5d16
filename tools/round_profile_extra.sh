#!/bin/bash
# The other BASELINE configurations on the GPU box (rank 64 / 13B / seq 4096, 70B widths, rank 32) -> gpurun_out/<tag>/; run after round_profile.sh.
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p $OUT
N="--no-cpu-baseline --no-traffic"
python bench.py --model 13b --rank 64 --seq 4096 --batch 2 --steps 10 $N > $OUT/bench_13b_r64_seq4096.json 2>> $OUT/bench.err
python bench.py --model 13b --rank 64 --seq 4096 --batch 2 --steps 10 --chains 2 $N > $OUT/bench_13b_r64_seq4096_chains2.json 2>> $OUT/bench.err
python bench.py --model 70b --steps 6 $N > $OUT/bench_70b.json 2>> $OUT/bench.err
python bench.py --model 70b --steps 6 --chains 2 $N > $OUT/bench_70b_chains2.json 2>> $OUT/bench.err
python bench.py --rank 32 --steps 20 $N > $OUT/bench_r32.json 2>> $OUT/bench.err
python bench.py --rank 32 --steps 20 --chains 1 $N > $OUT/bench_r32_chains1.json 2>> $OUT/bench.err
python bench.py --model 13b --steps 20 $N > $OUT/bench_13b_r16.json 2>> $OUT/bench.err
python bench.py --model 13b --rank 64 --seq 4096 --batch 2 --steps 10 --ablate all --no-cpu-baseline --no-traffic > $OUT/bench_13b_r64_seq4096_ablate.json 2>> $OUT/bench.err
for f in bench_13b_r64_seq4096 bench_13b_r64_seq4096_chains2 bench_70b bench_70b_chains2 bench_r32 bench_r32_chains1 bench_13b_r16 bench_13b_r64_seq4096_ablate; do python -c "import json; d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['adapter_hbm_roofline_frac'], 'chains', d['chains'])"; done
