#!/bin/bash
tag=${1:-r05e}
out=gpurun_out/$tag
mkdir -p $out
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["adapter_hbm_roofline_frac"], d["optimizer_in_backward"], "host", d["graph_replay_host_ms"])
except Exception as e:
    print('ERR', e)
PY
)"; }
for q in 4 8 16; do
export GPU_MAX_HW_QUEUES=$q
echo "== GPU_MAX_HW_QUEUES=$q"
run base_q$q --steps 40
run chains2_off_optin_q$q --steps 40 --chains 2 --defer-da off
run chains2_layer_q$q --steps 40 --chains 2 --defer-da layer
run chains2_bucket_q$q --steps 40 --chains 2 --defer-da bucket
run chains4_off_q$q --steps 40 --chains 4 --defer-da off --opt-in-backward off
run bucket1_q$q --steps 40 --defer-da bucket
done
