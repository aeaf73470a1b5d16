#!/bin/bash
# Round-end measurements on the GPU box -> gpurun_out/<tag>/ (copy what is judged into profiles/).  usage: round_profile.sh <tag>
TAG=${1:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
N="--no-cpu-baseline --no-traffic"
python bench.py > $OUT/bench.json 2> $OUT/bench.err                                      # the default command: 120 steps, cpu_baseline, traffic
python bench.py --steps 20 --warmup 5 $N > $OUT/bench_driver_command.json 2>> $OUT/bench.err   # what the driver runs (--steps 20 --warmup 5)
python bench.py --steps 20 --fuse-fwd off $N > $OUT/bench_unfused.json 2>> $OUT/bench.err      # the three-launch forward (round 3's)
python bench.py --steps 20 --force-comm $N > $OUT/bench_forcecomm.json 2>> $OUT/bench.err      # the N > 1 configuration on one GPU (one-rank RCCL)
python bench.py --steps 20 --graph bwd $N > $OUT/bench_graph_bwd.json 2>> $OUT/bench.err
python bench.py --steps 6 --graph off $N > $OUT/bench_graph_off.json 2>> $OUT/bench.err
python bench.py --steps 6 --variant vt $N > $OUT/bench_vt.json 2>> $OUT/bench.err
for b in 1 2 8; do python bench.py --steps 5 --batch $b $N > $OUT/bench_b$b.json 2>> $OUT/bench.err; done
python bench.py --steps 4 --dropout 0 $N > $OUT/bench_nodrop.json 2>> $OUT/bench.err
python bench.py --steps 4 --no-group $N > $OUT/bench_nogroup.json 2>> $OUT/bench.err
python bench.py --steps 6 --defer-da off $N > $OUT/bench_nodefer.json 2>> $OUT/bench.err
python bench.py --steps 20 --defer-da side --shadows-batch off $N > $OUT/bench_unbatched.json 2>> $OUT/bench.err   # one dA launch per unit, one shadows launch per unit (896 launches per step)
python bench.py --steps 3 --e2e $N > $OUT/bench_e2e.json 2>> $OUT/bench.err
tools/prof_run.sh $TAG > /dev/null 2>&1
for f in bench bench_driver_command bench_unfused bench_unbatched bench_forcecomm bench_graph_bwd bench_graph_off bench_vt bench_b1 bench_b2 bench_b8 bench_nodrop bench_nogroup bench_nodefer bench_e2e; do python - $OUT/$f.json $f <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d["value"], d["ms_per_step"], d.get("adapter_hbm_roofline_frac"), d.get("comm_exposed_ms"), d.get("end_to_end", ""))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
head -30 $OUT/kernel_trace.md
