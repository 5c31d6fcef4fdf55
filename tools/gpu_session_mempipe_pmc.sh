#!/bin/bash
O=gpurun_out/s8; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python tools/conv_probe.py --shapes 0,2,7 --tiles 3,5 --reps 4"
i=0
for set in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_LFIFO_STALL_CYCLES_sum" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex conv_igemm -f csv -d $R/$O/p$i -- bash -c "cd $R && $CMD" > $R/$O/p$i.log 2>&1)
  python tools/pmc_summary.py $O/p$i conv_igemm > $O/p$i.tsv 2>&1
done
find $O -name "*.csv" -size +4M -delete
