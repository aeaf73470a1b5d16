#!/bin/bash
# 13B r = 64 seq 4096 (BASELINE config 4): the family map (tools/r05_abl_families_r64.sh) shows the step as the SUM of its families -- the hub's dA / dB
# are not hidden at this rank.  Does a schedule without hub work (everything inside two chains) hide the exposed rank-space launches instead?
out=gpurun_out/abl; mkdir -p $out
r() { timeout 300 python bench.py --no-cpu-baseline --no-traffic "$@" 2>$out/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['adapter_hbm_roofline_frac'], 'chains', d['chains'], 'dA', d['defer_dA'], 'dB', d['defer_dB'], d['graph_topology'])" || tail -3 $out/err.txt; }
R64="--model 13b --rank 64 --seq 4096 --batch 2 --steps 16"
{
echo "default            $(r $R64)"
echo "c2                 $(r $R64 --chains 2)"
echo "c2 dA off          $(r $R64 --chains 2 --defer-da off)"
echo "c2 dA off dB off   $(r $R64 --chains 2 --defer-da off --defer-db off)"
echo "c2 dA main         $(r $R64 --chains 2 --defer-da main)"
echo "c1 dA off          $(r $R64 --chains 1 --defer-da off)"
echo "default            $(r $R64)"
} 2>&1 | tee $out/r64_nohub.txt
