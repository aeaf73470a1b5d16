#!/bin/bash
# the "marginal cost under the schedule" map of tools/r05_abl_families.sh for BASELINE config 4 (13B widths, r = 64, seq 4096, one chain) and for
# r = 32 (7B widths, two chains).  TIMING ONLY -- every ablation build leaves one kernel family's launches out and computes wrong results.
out=gpurun_out/abl; mkdir -p $out
r() { python bench.py --no-cpu-baseline --no-traffic "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['adapter_hbm_roofline_frac'], 'chains', d['chains'], d['entry_point_ms_per_pass'])"; }
R64="--model 13b --rank 64 --seq 4096 --batch 2 --steps 16"
r $R64 > /dev/null
{
echo "base       $(r $R64)"
for n in r64_xwm r64_crossfwd r64_yt r64_gy r64_dB r64_dA r64_dx; do echo "no_$n $(MOKA_HIP_LIB=$PWD/moka_amd/libmoka_hip_abl_$n.so r $R64)"; done
echo "no_crossbwd $(MOKA_HIP_LIB=$PWD/moka_amd/libmoka_hip_abl2.so r $R64)"
echo "no_dropout $(r $R64 --dropout 0)"
echo "no_optimizer $(r $R64 --no-optimizer)"
echo "base       $(r $R64)"
} 2>&1 | tee $out/families_r64.txt
R32="--rank 32 --steps 30"
{
echo "base     $(r $R32)"
for n in r64_xwm yx gs r64_dx dx da; do echo "no_$n $(MOKA_HIP_LIB=$PWD/moka_amd/libmoka_hip_abl_$n.so r $R32)"; done
echo "no_crossbwd $(MOKA_HIP_LIB=$PWD/moka_amd/libmoka_hip_abl2.so r $R32)"
echo "base     $(r $R32)"
echo "base c1  $(r $R32 --chains 1)"
for n in r64_xwm yx gs r64_dx dx da; do echo "c1 no_$n $(MOKA_HIP_LIB=$PWD/moka_amd/libmoka_hip_abl_$n.so r $R32 --chains 1)"; done
} 2>&1 | tee $out/families_r32.txt
