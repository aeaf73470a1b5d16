#!/bin/bash
# round 6, GPU call ah: the bench lines that changed with this session's kernels and launch rules, one box (copied to profiles/r06_*)
O=gpurun_out/r6ah; mkdir -p $O
N="--no-cpu-baseline --no-traffic"
b() { name=$1; shift; timeout 900 python bench.py "$@" > $O/$name.json 2>> $O/bench.err; }
b bench
b bench_driver_command --steps 20 --warmup 5 $N
b bench_r32 --steps 40 --rank 32 $N
b bench_r32_chains1 --steps 40 --rank 32 --chains 1 $N
b bench_13b_r64_seq4096 --model 13b --rank 64 --seq 4096 --batch 2 $N
b bench_13b_r64_seq4096_ablate --model 13b --rank 64 --seq 4096 --batch 2 --steps 10 --ablate all $N
b bench_13b_r64_seq4096_chains2 --model 13b --rank 64 --seq 4096 --batch 2 --chains 2 $N
b bench_13b_r16 --model 13b --steps 40 $N
b bench_70b --model 70b --steps 10 $N
for f in $O/bench*.json; do python - $f <<'PY'
import json, os, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(os.path.basename(sys.argv[1])[:-5], d["value"], d["ms_per_step"], d.get("adapter_hbm_roofline_frac"), "chains", d.get("chains"), (d.get("roofline") or {}).get("traffic"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
