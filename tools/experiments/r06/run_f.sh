#!/bin/bash
# round 6, GPU call f: the trainer path with checked captures
out=gpurun_out/r6f
mkdir -p $out
timeout 300 python -m pytest tests/test_gpu_trainer.py -x -q 2>&1 | tail -4
timeout 1100 python bench.py --e2e --steps 5 --no-cpu-baseline --ablate off > $out/e2e.json 2> $out/e2e.err; grep "e2e" $out/e2e.err
python -c "
import json
d=json.load(open('$out/e2e.json'))['end_to_end']; print(json.dumps({k:v for k,v in d.items() if k!='what'}, indent=1))"
