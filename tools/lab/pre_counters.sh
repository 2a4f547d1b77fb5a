#!/bin/bash
# Counter passes for k_preprocess_fwd vs k_preprocess_bwd (bound by neither VALU nor bandwidth -- name the stall).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pre; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
i=0
for C in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE" \
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES SQ_LEVEL_WAVES SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
 "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL GRBM_GUI_ACTIVE" \
 "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_ADDR_STALLED_BY_TD_CYCLES TA_TOTAL_WAVEFRONTS TA_FLAT_READ_WAVEFRONTS TA_FLAT_WRITE_WAVEFRONTS GRBM_GUI_ACTIVE" \
 "TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL TCC_EA0_WRREQ TCC_EA0_WRREQ_LEVEL TCC_HIT TCC_MISS TCC_REQ GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/c$i -- python $R/tools/profile_step.py --steps 3 > /tmp/c$i.log 2>&1
  python $R/tools/pmc_summary.py $(dirname $(find /tmp/c$i -name "*counter_collection.csv" | head -1)) --all | grep -A14 "k_preprocess" > $O/pass$i.txt
done
cat $O/pass*.txt
