#!/bin/bash
# Round 2, session 18: rocprofv3 kernel stats of the pull_item image path (jpeg_idct_k / jpeg_color_k / fast_base_transform).
O=gpurun_out/r2s18; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/stats -- bash -c "cd $R && python tools/jpeg_probe.py" > $R/$O/stats.log 2>&1)
tail -3 $O/stats.log | cut -c1-300
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r2s18/stats/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('jpeg', 'fast_base', 'copy')):
        print('%-90s calls %5s avg %8.1f ns' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])))
PY
