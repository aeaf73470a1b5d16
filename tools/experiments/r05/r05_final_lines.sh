#!/bin/bash
# the bench lines of the round's profile set, final build, ONE box, one call -> gpurun_out/r05/ (tools/round_collect.py r05 copies them)
OUT=gpurun_out/r05; mkdir -p $OUT
N="--no-cpu-baseline --no-traffic"
b() { name=$1; shift; timeout 400 python bench.py "$@" > $OUT/$name.json 2>> $OUT/bench.err; }
b bench
b bench_driver_command --steps 20 --warmup 5 --no-cpu-baseline
b bench_chains1 --chains 1 --steps 60 --no-cpu-baseline
b bench_vt --variant vt --steps 40 $N
b bench_r32 --rank 32 --steps 20 $N
b bench_r32_chains1 --rank 32 --chains 1 --steps 20 $N
for n in 1 2 3 8; do b bench_b$n --batch $n --steps 30 $N; done
for n in 2 4 8; do b bench_b${n}_chains1 --batch $n --chains 1 --steps 30 $N; done
b bench_forcecomm --steps 40 --force-comm $N
b bench_forcecomm_bf16 --steps 40 --force-comm --comm-bf16 $N
b bench_forcecomm_chains1 --steps 40 --force-comm --chains 1 $N
b bench_nodrop --dropout 0 --steps 40 $N
b bench_noopt --no-optimizer --steps 40 $N
b bench_70b --model 70b --steps 6 $N
b bench_13b_r64_seq4096 --model 13b --rank 64 --seq 4096 --batch 2 --steps 10 $N
b bench_13b_r16 --model 13b --steps 10 $N
for f in $OUT/bench*.json; do python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'[15:], d['value'], d['ms_per_step'], d['adapter_hbm_roofline_frac'], d['chains'], d['roofline']['traffic'], d['roofline']['frac'], d['comm_exposed_ms'])" 2>/dev/null; done
