#!/bin/bash
# round 6, GPU call d: ABI 0.7.0 (moka_opts.struct_size / seed_dev) through the whole GPU suite, the default line, the dx marginal,
# token-class chains at B = 1..4, the trainer path (--e2e)
out=gpurun_out/r6d
mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; tail -5 $out/gpu_tests.log
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d["ms_per_step"], d["adapter_hbm_roofline_frac"], "chains", d["chains"], d.get("ablation", {}).get("families"), d["roofline"].get("in_schedule") and d["roofline"]["in_schedule"]["marginal_ms"])
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
timeout 300 python bench.py --steps 20 > $out/bench_default.json 2> $out/bench_default.err; line $out/bench_default.json
timeout 300 python bench.py --steps 20 --seed-dev off --no-cpu-baseline --ablate off > $out/bench_seedoff.json 2> $out/bench_seedoff.err; line $out/bench_seedoff.json
timeout 400 python bench.py --steps 20 --ablate dx,none --no-cpu-baseline > $out/bench_abl_dx.json 2> $out/bench_abl_dx.err; line $out/bench_abl_dx.json
for b in 4 1 2 3; do
  timeout 300 python bench.py --batch $b --steps 20 --no-cpu-baseline --ablate off > $out/b${b}_sample.json 2> $out/b${b}_sample.err; line $out/b${b}_sample.json
  timeout 300 python bench.py --batch $b --chains 2 --chain-split class --steps 20 --no-cpu-baseline --ablate off > $out/b${b}_class.json 2> $out/b${b}_class.err; line $out/b${b}_class.json
done
timeout 900 python bench.py --e2e --steps 5 --no-cpu-baseline --ablate off > $out/e2e.json 2> $out/e2e.err; grep "e2e" $out/e2e.err
