#!/bin/bash
# Final measurement pass of the round: full GPU test suite, smoke, bench (default + batch 1), single-stream rocprof stats.
O=gpurun_out/final9; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
export YOLACT_AMD_TUNE_CACHE=$PWD/$O/tune.json
timeout 600 python bench.py --layers > $O/bench.json 2> $O/bench_layers.txt; cut -c1-420 $O/bench.json
timeout 300 python bench.py --batch 1 --steps 50 --no-cpu-baseline > $O/bench_b1.json 2> /dev/null; cut -c100-230 $O/bench_b1.json
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/stats2 -- bash -c "cd $R && python bench.py --steps 10 --warmup 2 --no-cpu-baseline" > $R/$O/stats2.log 2>&1)
(cd /tmp && YOLACT_AMD_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/stats1 -- bash -c "cd $R && python bench.py --steps 10 --warmup 2 --no-cpu-baseline" > $R/$O/stats1.log 2>&1)
for m in 0 2 1; do YOLACT_AMD_WINOGRAD=$m timeout 300 python tools/e2e_error.py 2>/dev/null; done > $O/e2e_error.txt; cat $O/e2e_error.txt
