#!/bin/bash
# One gpurun call: sanity tests, baseline bench, microbench peaks, ablation probe, PMC passes. Outputs under gpurun_out/s1/.
set -x
O=gpurun_out/s1; mkdir -p $O
export TMPDIR=/tmp YOLACT_AMD_TUNE_CACHE=$PWD/$O/tune.json
tools/microbench.bin > $O/microbench.json 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python bench.py --layers > $O/bench.json 2> $O/bench_layers.txt
timeout 600 python tools/conv_probe.py --ablate 0,1,3,7 --tiles 1,3 --reps 10 > $O/probe.txt 2>&1
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE GRBM_COUNT --kernel-include-regex conv_igemm -f csv -d $GRAFT_REPO_ROOT/$O/pmc1 -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $GRAFT_REPO_ROOT/$O/pmc1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES SQ_INSTS_VALU --kernel-include-regex conv_igemm -f csv -d $GRAFT_REPO_ROOT/$O/pmc2 -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $GRAFT_REPO_ROOT/$O/pmc2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex conv_igemm -f csv -d $GRAFT_REPO_ROOT/$O/pmc3 -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $GRAFT_REPO_ROOT/$O/pmc3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-include-regex conv_igemm -f csv -d $GRAFT_REPO_ROOT/$O/pmc4 -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $GRAFT_REPO_ROOT/$O/pmc4.log 2>&1
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do python tools/pmc_summary.py $O/pmc$i conv_igemm > $O/pmc$i.tsv 2>&1; done
# keep the merged-back payload small
find $O -name "*.csv" -size +8M -delete
du -sh $O
