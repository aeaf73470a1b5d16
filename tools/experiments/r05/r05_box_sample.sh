#!/bin/bash
# one sample of the headline on whatever box the call lands on -> gpurun_out/boxes/<unique id>.json (value, ms, box id, clocks)   usage: r05_box_sample.sh
mkdir -p gpurun_out/boxes
ID=$(rocm-smi --showuniqueid 2>/dev/null | grep -o "0x[0-9a-f]*" | head -1); ID=${ID:-$(hostname)}
CLK=$(rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | tr -s ' ' | tr '\n' ';')
for k in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'box': '$ID', 'run': $k, 'command': '--steps 20 --warmup 5', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'frac': d['adapter_hbm_roofline_frac'], 'traffic': d['roofline']['traffic'], 'clocks': '''$CLK'''}))
" | tee -a gpurun_out/boxes/$ID.jsonl
done
