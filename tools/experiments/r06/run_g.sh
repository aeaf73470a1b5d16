#!/bin/bash
# round 6, GPU call g: the split library through the whole GPU suite, the PMC traffic passes of THIS source, default line, trainer path
out=gpurun_out/r6g
mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; tail -4 $out/gpu_tests.log
timeout 900 bash tools/pmc_traffic.sh r6g_pmc > $out/pmc.log 2>&1; tail -12 $out/pmc.log
timeout 300 python bench.py --steps 20 --no-traffic > $out/bench_default.json 2> $out/bench_default.err
python -c "
import json
d=json.load(open('$out/bench_default.json')); print(d['ms_per_step'], d['adapter_hbm_roofline_frac'], d['roofline']['in_schedule'])"
timeout 900 python bench.py --e2e --steps 5 --no-cpu-baseline --ablate off > $out/e2e.json 2> $out/e2e.err; grep "e2e:" $out/e2e.err
python -c "
import json
d=json.load(open('$out/e2e.json'))['end_to_end']; print({k:v for k,v in d.items() if k not in ('what','captures','live')})"
