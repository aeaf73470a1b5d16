#!/bin/bash
out=gpurun_out/r05bk; mkdir -p $out
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-traffic "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["adapter_hbm_roofline_frac"], "chains", d["chains"], d["graph"], d["comm_exposed_ms"], d["distributed"]["bucket_layers"])
except Exception as e:
    print('ERR', e); print(open('$out/$name.err').read()[-600:])
PY
)"; }
run default --steps 40
for nb in 8 6 5 4 3 2; do run fc_b$nb --steps 40 --force-comm --buckets $nb; done
run fc_b8_again --steps 40 --force-comm
run fc_b4_tail2 --steps 40 --force-comm --buckets 4 --tail-layers 2
run fc_b4_bf16 --steps 40 --force-comm --buckets 4 --comm-bf16
run bwd_b4 --steps 40 --graph bwd --buckets 4
