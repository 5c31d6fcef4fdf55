#!/bin/bash
# PMC HBM-traffic passes + kernel-trace stats of the bench command (single-stream so kernels are serialised).
O=gpurun_out/traffic; mkdir -p $O
export TMPDIR=/tmp YOLACT_AMD_TUNE_CACHE=$PWD/$O/tune.json YOLACT_AMD_STREAMS=1
R=$GRAFT_REPO_ROOT
cp profiles/r01_tune_v9.json $O/tune.json      # the tile / Winograd choices of the committed bench run
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex conv_igemm -f csv -d $R/$O/fetch -- bash -c "cd $R && $CMD" > $R/$O/fetch.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex conv_igemm -f csv -d $R/$O/write -- bash -c "cd $R && $CMD" > $R/$O/write.log 2>&1)
python tools/traffic_summary.py $O/fetch $O/write > $O/r01_traffic.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/stats -- bash -c "cd $R && python bench.py --steps 10 --warmup 2 --no-cpu-baseline" > $R/$O/stats.log 2>&1)
find $O -name "*counter_collection.csv" -size +6M -delete
cat $O/r01_traffic.json | head -30
