#!/bin/bash
# Round-end measurements on the GPU box -> gpurun_out/<tag>/ (copy what is judged into profiles/ with tools/round_collect.py).  usage: round_profile.sh <tag>
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p $OUT
N="--no-cpu-baseline --no-traffic"
b() { name=$1; shift; timeout 900 python bench.py "$@" > $OUT/$name.json 2>> $OUT/bench.err; }
b bench                                                        # the default command: 120 steps, cpu_baseline, traffic (2 part-batch chains, hub-shaped graph)
b bench_driver_command --steps 20 --warmup 5 $N                # what the driver runs (--steps 20 --warmup 5)
b bench_chains1 --steps 60 --chains 1 $N                       # one chain: round 4's graph shape with the chain-first capture order
b bench_chains1_sidefirst --steps 60 --chains 1 --capture-order side-first $N        # ... and exactly as round 4 captured it
b bench_chains1_probe --steps 20 --chains 1 --probe-forward $N  # + the forward alone: hipGraph replay vs live launches (HIP events, no profiler)
b bench_defer_layer --steps 60 --defer-da layer $N              # 2 chains, a layer's dA_m as one launch on the hub
b bench_defer_side --steps 60 --defer-da side $N                # 2 chains, per-unit dA_m launches at the layer's end
b bench_nodefer --steps 60 --defer-da off $N                    # 2 chains, dA_m inside the chains
b bench_chains4 --steps 20 --chains 4 $N                        # (more lists than the executor runs side by side)
b bench_forcecomm --steps 40 --force-comm $N                    # the N > 1 configuration on one GPU (one-rank RCCL): per-bucket graphs, 2 chains
b bench_forcecomm_bf16 --steps 40 --force-comm --comm-bf16 $N   # ... with the bf16 payload (optimizer slices behind the widened sum)
b bench_forcecomm_chains1 --steps 40 --force-comm --chains 1 $N
b bench_graph_bwd --steps 40 --graph bwd $N
b bench_graph_off --steps 6 --graph off $N
b bench_vt --steps 40 --variant vt $N
for n in 1 2 3 8; do b bench_b$n --steps 40 --batch $n $N; done
for n in 2 4 8; do b bench_b${n}_chains1 --steps 30 --batch $n --chains 1 $N; done
b bench_nodrop --steps 40 --dropout 0 $N
b bench_noopt --steps 40 --no-optimizer $N
b bench_nogroup --steps 20 --no-group $N
b bench_verify --steps 3 --no-optimizer --verify-graph $N      # the captured schedule == the live launches (graph_check)
b bench_verify_opt --steps 3 --verify-graph $N                 # ... and WITH the optimizer: 3 steps as the graph vs 3 steps live from identical state
b bench_ablate --steps 20 --ablate all --no-cpu-baseline       # in-schedule marginal of every kernel family (ScheduleConfig.skip: no rebuilt library)
for n in 1 2 3 4; do b bench_b${n}_class --steps 40 --batch $n --chains 2 --chain-split class $N; done    # chains INSIDE the samples (by token class)
b bench_seed_off --steps 40 --seed-dev off $N                  # the dropout epoch in device memory (fresh masks per replay): what it costs
b bench_e2e --steps 3 --e2e --ablate off $N                    # the trainer path: decoder stack + attach, live and as a captured GraphedTrainStep
tools/prof_run.sh $TAG > /dev/null 2>&1
for f in $OUT/bench*.json; do python - $f <<'PY'
import json, os, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(os.path.basename(sys.argv[1])[:-5], d["value"], d["ms_per_step"], d.get("adapter_hbm_roofline_frac"), "chains", d.get("chains"), d.get("defer_dA"), d.get("comm_exposed_ms"),
          d.get("forward_only") or "", d.get("graph_check") or "", d.get("end_to_end", ""))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
head -30 $OUT/kernel_trace.md
