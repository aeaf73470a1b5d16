#!/bin/bash
# Stall / memory-pipeline counter passes of a short bench run (one rocprofv3 --pmc pass per group) -> gpurun_out/<tag>/summary.md
# usage: pmc_stall.sh <tag> ["<bench args>"]      (run on the GPU box from the repo root)
TAG=${1:-stall}; BARGS=${2:---layers 2}
exec tools/pmc_groups.sh $TAG "$BARGS" \
  "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
  "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
  "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
  "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_TAG_STALL_sum" \
  "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVES_EQ_64 SQ_LEVEL_WAVES"
