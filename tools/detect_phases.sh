#!/bin/bash
# per-kernel times of Detect with the K2 ablations (diagnostics build only): rocprofv3 kernel stats, one process per setting
O=gpurun_out/$1; mkdir -p $O; R=$(pwd)
for B in 1 8; do for A in ${ABLS:-0 1 4 5}; do
  (cd /tmp && export TMPDIR=/tmp && PROBE_BATCH=$B PROBE_ABL=$A timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/d_${B}_$A -- bash -c "cd $R && python tools/detect_probe.py" > $R/$O/d_${B}_$A.log 2>&1)
  f=$(find $O/d_${B}_$A -name "*kernel_stats.csv" | head -1)
  echo "batch $B ablate $A: $(grep -h 'ablate=' $O/d_${B}_$A.log | cut -c1-60)"
  grep -E "softmax_keep|class_topk|final_topk" $f | awk -F'","' '{printf "   %-40s calls %s avg %.1f us\n", substr($1,2,40), $2, $4/1000}'
done; done | tee $O/detect_phases.txt
