#!/bin/bash
# Round 2, session 13: final lock-in (padded class rows, hoisted segment stores, peeled x3 loop): tune table (split + exact-fp32 keys), full GPU suite, smoke, PMC traffic,
# bench (default + other BASELINE configs), rocprofv3 kernel stats.
O=gpurun_out/r2s13; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python tools/make_tune_table.py --fresh > $O/tune.log 2>&1
YOLACT_AMD_SPLIT=0 timeout 900 python tools/make_tune_table.py --only configs1_r50_b8 r50_b1 r50_b2 --copy-to $O/tune/gfx950.json >> $O/tune.log 2>&1; grep -E "plan|table" $O/tune.log | cut -c1-200
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -rA > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
# HBM traffic of the conv kernels (PMC, separate passes, single stream)
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
(cd /tmp && YOLACT_AMD_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex conv_igemm -f csv -d $R/$O/fetch -- bash -c "cd $R && $CMD" > $R/$O/fetch.log 2>&1)
(cd /tmp && YOLACT_AMD_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex conv_igemm -f csv -d $R/$O/write -- bash -c "cd $R && $CMD" > $R/$O/write.log 2>&1)
python tools/traffic_summary.py $O/fetch $O/write > $O/r02_traffic.json 2> $O/traffic.err; cp $O/r02_traffic.json profiles/r02_traffic.json; head -c 600 $O/r02_traffic.json
find $O -name "*counter_collection.csv" -size +4M -delete
timeout 900 python bench.py --layers > $O/bench.json 2> $O/bench_layers.txt; head -1 $O/bench.json | cut -c1-420
timeout 300 python bench.py --batch 1 --steps 50 --no-cpu-baseline --no-secondary > $O/bench_b1.json 2>/dev/null; head -1 $O/bench_b1.json | cut -c90-230
timeout 400 python bench.py --config yolact_base_config --batch 16 --steps 10 --no-cpu-baseline --no-secondary > $O/bench_r101_b16.json 2>/dev/null; head -1 $O/bench_r101_b16.json | cut -c90-240
timeout 400 python bench.py --config yolact_im700_config --batch 8 --steps 10 --no-cpu-baseline --no-secondary > $O/bench_im700_b8.json 2>/dev/null; head -1 $O/bench_im700_b8.json | cut -c90-240
timeout 400 python bench.py --config yolact_plus_resnet50_config --batch 8 --steps 10 --no-cpu-baseline --no-secondary > $O/bench_plus_b8.json 2>/dev/null; head -1 $O/bench_plus_b8.json | cut -c90-240
timeout 400 python bench.py --config yolact_darknet53_config --batch 8 --steps 10 --no-cpu-baseline --no-secondary > $O/bench_darknet_b8.json 2>/dev/null; head -1 $O/bench_darknet_b8.json | cut -c90-240
YOLACT_AMD_SPLIT=0 timeout 400 python bench.py --no-cpu-baseline --no-secondary > $O/bench_fp32only.json 2>/dev/null; head -1 $O/bench_fp32only.json | cut -c90-240
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/stats2 -- bash -c "cd $R && python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary" > $R/$O/stats2.log 2>&1)
(cd /tmp && YOLACT_AMD_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/stats1 -- bash -c "cd $R && python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary" > $R/$O/stats1.log 2>&1)
(cd /tmp && YOLACT_AMD_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/stats_post -- bash -c "cd $R && python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --with-postprocess" > $R/$O/stats_post.log 2>&1)
ls $O
