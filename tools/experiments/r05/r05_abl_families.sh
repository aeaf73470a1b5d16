#!/bin/bash
# same-box "marginal cost under overlap" map (TIMING ONLY: every ablation build computes wrong results): the default step with ONE kernel
# family's launches removed from the library -- what the step gains is what that family costs inside the two-chain schedule, to be read
# against its stand-alone time in profiles/r05_kernel_trace.md.  Builds: sed-edited copies of the kernel source (one hipLaunchKernelGGL
# line commented out each; tools/README.md).   usage: r05_abl_families.sh [bench args]
out=gpurun_out/abl; mkdir -p $out
r() { python bench.py --no-cpu-baseline --no-traffic --steps 60 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['adapter_hbm_roofline_frac'], 'chains', d['chains'])"; }
r "$@" > /dev/null
for i in 1 2; do
  echo "base  $(r "$@")"
  for n in xs yx gs dx da; do echo "no_$n $(MOKA_HIP_LIB=$PWD/moka_amd/libmoka_hip_abl_$n.so r "$@")"; done
  echo "no_keys  $(MOKA_HIP_LIB=$PWD/moka_amd/libmoka_hip_abl.so r "$@")"
  echo "no_cross $(MOKA_HIP_LIB=$PWD/moka_amd/libmoka_hip_abl2.so r "$@")"
  echo "no_dropout $(r --dropout 0 "$@")"
  echo "no_optimizer $(r --no-optimizer "$@")"
done 2>&1 | tee $out/families$TAG.txt
