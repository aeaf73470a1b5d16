#!/bin/bash
tag=${1:-r05q}
out=gpurun_out/$tag
mkdir -p $out
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["adapter_hbm_roofline_frac"], "chains", d["chains"], d["graph"], "host", d["graph_replay_host_ms"])
except Exception as e:
    print('ERR', e); print(open('$out/$name.err').read()[-600:])
PY
)"; }
run default --steps 40
for gq in 2 3 6 8 12; do
for hq in 4 8; do
export DEBUG_HIP_FORCE_GRAPH_QUEUES=$gq GPU_MAX_HW_QUEUES=$hq
echo "== graph queues $gq hw queues $hq"
run c2_g${gq}_h${hq} --steps 30 --chains 2
run c4_g${gq}_h${hq} --steps 30 --chains 4
run c1_g${gq}_h${hq} --steps 30 --chains 1
done
done
unset DEBUG_HIP_FORCE_GRAPH_QUEUES GPU_MAX_HW_QUEUES
run default2 --steps 40
