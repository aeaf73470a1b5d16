#!/bin/bash
# MFMA / LDS counters of a short bench run (4 layers), one counter group per pass -> gpurun_out/<tag>/summary.md.
# usage: pmc_mfma.sh [tag] ["<bench args>"]      (run on the GPU box from the repo root)
REPO=$PWD
OUT=$REPO/gpurun_out/${1:-pmc_mfma}
BARGS=${2:---layers 4}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcm_*
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmcm_$i -o p$i -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-traffic --graph off $BARGS > $OUT/run$i.log 2>&1 || echo "pass $i ($grp) failed" >> $OUT/errors.txt
done
python $REPO/tools/rocpd_pmc_multi.py $(find /tmp/pmcm_* -name "*.db") > $OUT/summary.md 2>> $OUT/errors.txt
tail -5 $OUT/errors.txt 2>/dev/null
