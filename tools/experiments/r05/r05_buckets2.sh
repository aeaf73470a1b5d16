#!/bin/bash
out=gpurun_out/r05bk2; mkdir -p $out
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-traffic "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$? $(python - <<PY
import json
try:
    d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["adapter_hbm_roofline_frac"], "chains", d["chains"], d["graph"], d["comm_exposed_ms"], d["distributed"]["bucket_layers"], d["distributed"]["last_bucket_bytes"], d["optimizer_in_backward"])
except Exception as e:
    print('ERR', e); print(open('$out/$name.err').read()[-600:])
PY
)"; }
run default --steps 40
run fc --steps 40 --force-comm
run fc_bf16 --steps 40 --force-comm --comm-bf16
run fc_b8 --steps 40 --force-comm --buckets 8 --tail-layers 1
run fc_chains1 --steps 40 --force-comm --chains 1
run fc2 --steps 40 --force-comm
MOKA_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --layers 4 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic 2>&1 | tail -1 | cut -c1-300
python -m pytest tests/test_gpu_dp.py tests/test_gpu_bench_graph.py -q 2>&1 | tail -2
