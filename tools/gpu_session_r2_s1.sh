#!/bin/bash
# Round 2, GPU session 1: inner-loop probe (fp32 MFMA vs bf16x3 split), tune table, full GPU test suite, smoke, bench, rocprof.
O=gpurun_out/r2s1; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/split_probe.hip -o /tmp/split_probe.bin && timeout 300 /tmp/split_probe.bin ) > $O/split_probe.json 2> $O/split_probe.err; cat $O/split_probe.json
timeout 1500 python tools/make_tune_table.py --fresh --copy-to $O/tune/gfx950.json > $O/tune.log 2>&1; tail -14 $O/tune.log
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -rA > $O/pytest.log 2>&1; tail -30 $O/pytest.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py --layers > $O/bench.json 2> $O/bench_layers.txt; cut -c1-600 $O/bench.json; tail -5 $O/bench_layers.txt
(cd /tmp && YOLACT_AMD_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/stats1 -- bash -c "cd $R && python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary" > $R/$O/stats1.log 2>&1)
ls $O/stats1 | head
