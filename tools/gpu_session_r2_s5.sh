#!/bin/bash
# Round 2, session 5: bf16x3 with pre-split filter planes (PREC 2): unit parity, per-layer probe (planes vs on-the-fly), bench.
O=gpurun_out/r2s5; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -k "bf16x3 or conv_plain or winograd_matches" -x > $O/pytest_x3.log 2>&1; tail -3 $O/pytest_x3.log | cut -c1-300
PROBE_PLANES=0 timeout 200 python tools/conv_probe.py --shapes 0,1,4,5 --tiles 37,33,35 --reps 4 2>&1 | grep -v amdgpu > $O/probe_fly.txt
PROBE_PLANES=1 timeout 200 python tools/conv_probe.py --shapes 0,1,4,5 --tiles 37,33,35,51,53 --reps 4 2>&1 | grep -v amdgpu > $O/probe_planes.txt
paste $O/probe_fly.txt $O/probe_planes.txt | cut -c1-200
export YOLACT_AMD_SPLIT=1
export YOLACT_AMD_TUNE_CACHE=$PWD/$O/tune_x3.json
timeout 900 python bench.py --layers --no-cpu-baseline > $O/bench_x3.json 2> $O/bench_x3_layers.txt; head -1 $O/bench_x3.json | cut -c1-330
grep -vE "^tune|^wino|amdgpu|socket" $O/bench_x3_layers.txt | cut -c1-130
