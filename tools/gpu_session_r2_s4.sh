#!/bin/bash
# Round 2, session 4: where do the bf16x3 tiles stall?  PMC passes (separate runs, --pmc with --kernel-trace only) + block traces.
O=gpurun_out/r2s4; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python tools/conv_probe.py --shapes 0,1,4,5 --tiles 5,37,33,35 --reps 4"
timeout 200 $CMD > $O/probe_times.txt 2>&1; cat $O/probe_times.txt
i=0
for set in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex conv_igemm -f csv -d $R/$O/p$i -- bash -c "cd $R && $CMD" > $R/$O/p$i.log 2>&1)
  python tools/pmc_summary.py $O/p$i conv_igemm > $O/p$i.tsv 2>&1
done
find $O -name "*.csv" -size +2M -delete
for st in "1 37" "1 5" "5 35" "5 3" "0 33"; do timeout 120 python tools/conv_trace.py $st 2>&1 | grep -v "^ticks\|^start\|^end\|^epi.math\|^epi.store\|amdgpu.ids"; echo; done > $O/traces.txt
cat $O/traces.txt | head -70
