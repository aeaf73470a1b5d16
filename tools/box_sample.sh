#!/bin/bash
# The driver's command twice on whatever box this gpurun call landed on -> one JSON line per run appended to gpurun_out/boxes/<tag>_bench_boxes.jsonl
# (uid = rocm-smi's unique id of the GPU).  usage: box_sample.sh [tag]      -- run it in several gpurun calls, then copy the file to profiles/
TAG=${1:-r06}
mkdir -p gpurun_out/boxes
UID_=$(rocm-smi --showuniqueid 2>/dev/null | grep -o "0x[0-9a-f]*" | head -1)
for i in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --ablate off 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'uid': '$UID_', 'run': $i, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'frac': d['adapter_hbm_roofline_frac'], 'chains': d['chains'], 'graph': d['graph']}))" | tee -a gpurun_out/boxes/${TAG}_bench_boxes.jsonl
done
