#!/bin/bash
# Round 2, session 7: pointwise loader + stem on x3: kernel tests, bench (SPLIT=1).
O=gpurun_out/r2s7; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -x > $O/pytest_kernels.log 2>&1; tail -3 $O/pytest_kernels.log | cut -c1-300
export YOLACT_AMD_SPLIT=1
export YOLACT_AMD_TUNE_CACHE=$PWD/$O/tune_x3.json
timeout 900 python bench.py --layers --no-cpu-baseline > $O/bench_x3.json 2> $O/bench_x3_layers.txt; head -1 $O/bench_x3.json | cut -c1-330
grep -vE "^tune|^wino|amdgpu|socket" $O/bench_x3_layers.txt | cut -c1-130 | head -30
