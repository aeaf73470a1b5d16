#!/bin/bash
# round 6, GPU call v: do the re-reads of the weight fragments miss L2 because a column range's workgroups are spread over the XCDs?  moka_dxgt_kernel<64, G> with
# 8 column ranges (dx_group=3: workgroup (x, y) has linear id 8 y + x -> XCD x, every XCD stages ONE range's weights) against the default 10 (ranges spread over 4 XCDs each)
TAG=r6v; REPO=$PWD; mkdir -p gpurun_out/$TAG
ARGS="--model 13b --rank 64 --seq 4096 --batch 2 --layers 2 --steps 1 --warmup 1 --graph off --no-cpu-baseline --no-traffic --defer-da off --ablate off"
cd /tmp && export TMPDIR=/tmp
for t in 0 3 5; do
rm -rf /tmp/pmcF
MOKA_HIP_LIB=$REPO/moka_amd/libmoka_hip_diag.so MOKA_TUNE=dx_group=$t timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmcF -o f -- python $REPO/bench.py $ARGS > $REPO/gpurun_out/$TAG/runF_$t.log 2>&1
F=$(find /tmp/pmcF -name "*.db" | head -1)
echo "== dx_group=$t"; python $REPO/tools/rocpd_pmc_summary.py $F FETCH_SIZE | grep "dxgt\|dxt" | tail -3
done 2>&1 | tee $REPO/gpurun_out/$TAG/fetch.txt
