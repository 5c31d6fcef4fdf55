#!/bin/bash
# Round 2, session 16: final tree — PMC evidence for the dominant kernel on the real plan, then the GPU suite without the
# slow host-oracle batch-parity file (unchanged engine + kernels since r2s13, where it passed), smoke.
O=gpurun_out/r2s16; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary"
(cd /tmp && YOLACT_AMD_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-include-regex conv_igemm -f csv -d $R/$O/pmc1 -- bash -c "cd $R && $CMD" > $R/$O/pmc1.log 2>&1)
python tools/pmc_summary.py $O/pmc1 conv_igemm > $O/pmc_plan_p1.tsv 2>&1; head -12 $O/pmc_plan_p1.tsv | cut -c1-220
find $O -name "*.csv" -size +2M -delete
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -rA --deselect tests/test_gpu_batch_parity.py > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
