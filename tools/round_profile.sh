#!/bin/bash
# Round-end measurements on the GPU box -> gpurun_out/<tag>/ (copy what is judged into profiles/).  usage: round_profile.sh <tag>
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py --steps 10 > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 6 --graph off --no-cpu-baseline > $OUT/bench_graph_off.json 2>> $OUT/bench.err
python bench.py --steps 6 --variant vt --no-cpu-baseline --no-traffic > $OUT/bench_vt.json 2>> $OUT/bench.err
for b in 1 2 8; do python bench.py --steps 5 --batch $b --no-cpu-baseline --no-traffic > $OUT/bench_b$b.json 2>> $OUT/bench.err; done
python bench.py --steps 4 --dropout 0 --no-cpu-baseline --no-traffic > $OUT/bench_nodrop.json 2>> $OUT/bench.err
python bench.py --steps 4 --no-group --no-cpu-baseline --no-traffic > $OUT/bench_nogroup.json 2>> $OUT/bench.err
python bench.py --steps 6 --defer-da off --no-cpu-baseline --no-traffic > $OUT/bench_nodefer.json 2>> $OUT/bench.err
python bench.py --steps 3 --e2e --no-cpu-baseline --no-traffic > $OUT/bench_e2e.json 2>> $OUT/bench.err
tools/prof_run.sh $TAG > /dev/null 2>&1
for f in bench bench_graph_off bench_vt bench_b1 bench_b2 bench_b8 bench_nodrop bench_nogroup bench_nodefer bench_e2e; do python - $OUT/$f.json $f <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2], d["value"], d["ms_per_step"], d.get("adapter_hbm_roofline_frac"), d.get("end_to_end", ""))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
head -30 $OUT/kernel_trace.md
