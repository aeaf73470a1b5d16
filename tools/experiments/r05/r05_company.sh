#!/bin/bash
out=gpurun_out/r05co; mkdir -p $out
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-traffic "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["adapter_hbm_roofline_frac"], "chains", d["chains"])
except Exception as e:
    print('ERR', e); print(open('$out/$name.err').read()[-600:])
PY
)"; }
run r32_on --rank 32 --steps 20
run r32_off --rank 32 --steps 20 --company-hint off
run r32_on2 --rank 32 --steps 20
run r32_off2 --rank 32 --steps 20 --company-hint off
run r16_on --steps 60
run r16_off --steps 60 --company-hint off
run r16_on2 --steps 60
run r16_off2 --steps 60 --company-hint off
run vt_on --steps 60 --variant vt
run vt_off --steps 60 --variant vt --company-hint off
run b8_on --steps 30 --batch 8
run b8_off --steps 30 --batch 8 --company-hint off
