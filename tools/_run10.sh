mkdir -p gpurun_out/r4j
python -c "import torch; print(torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
for pr in 1 -1; do
MOKA_SIDE_PRIORITY=$pr timeout 900 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --e2e > gpurun_out/r4j/e2e_p$pr.json 2> gpurun_out/r4j/e2e_p$pr.err
python - $pr <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r4j/e2e_p%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['end_to_end'])
PY
done
