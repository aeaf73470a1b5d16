#!/bin/bash
# round 6, GPU call m: workgroups per CU of the rank-pad-64 chunk-walk down-projection (moka_tune xa_ng -> slices per projection), per group size:
# moka_xwm_kernel<64, false, 3> holds 208 registers and 96 KB of LDS (ONE workgroup per CU), yet runs on the 3-per-CU rule of the single projection
# (640 workgroups of 2 chunks at 8192 x 5120: 2.5 rounds).  Kernel trace of 8 layers, everything in one chain, diagnostics library.
TAG=r6m; REPO=$PWD; mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
for ng in 3 1 2 4; do
rm -rf /tmp/prof_$TAG
MOKA_HIP_LIB=$REPO/moka_amd/libmoka_hip_diag.so MOKA_TUNE=xa_ng=$ng timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o r -- python $REPO/bench.py --model 13b --rank 64 --seq 4096 --batch 2 --layers 8 --steps 3 --warmup 2 --no-cpu-baseline --no-traffic --ablate off --defer-da off --graph off --chains 1 > $REPO/gpurun_out/$TAG/prof_run_$ng.log 2>&1
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
echo "== xa_ng=$ng"; python $REPO/tools/rocpd_summary.py $DB bygrid 0.5 1.0 | grep "xwm_kernel<64, false\|cross_fwd" 
done 2>&1 | tee $REPO/gpurun_out/$TAG/xa_ng.txt
