#!/bin/bash
# Round 2, session 15 (A/B): WN = 1 tiles (128x64n, 128x128n, 256x128n8, 128x256w8x3) + packed subtractions in the bf16x3
# split: kernel tests, then a forced re-tune bench against the shipped-table bench of the same binary.
O=gpurun_out/r2s15; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -x > $O/pytest_kernels.log 2>&1; tail -3 $O/pytest_kernels.log | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-secondary > $O/bench_table.json 2> /dev/null; head -1 $O/bench_table.json | cut -c90-240
export YOLACT_AMD_AUTOTUNE=force
export YOLACT_AMD_TUNE_CACHE=$PWD/$O/tune.json
timeout 900 python bench.py --layers --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/bench_layers.txt; head -1 $O/bench.json | cut -c90-240
grep -E "^tune|^wino" $O/bench_layers.txt | grep -E "n8?x3|128x256w8x3| 128x64n| 128x128n| 256x128n8" | cut -c1-60 | head -40
