#!/bin/bash
# Round 2, session 11: peeled / unrolled bf16x3 main loop: kernel tests + forced re-tune bench.
O=gpurun_out/r2s11; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -x > $O/pytest_kernels.log 2>&1; tail -3 $O/pytest_kernels.log | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline --no-secondary > $O/bench_table.json 2> /dev/null; head -1 $O/bench_table.json | cut -c90-240
export YOLACT_AMD_AUTOTUNE=force
export YOLACT_AMD_TUNE_CACHE=$PWD/$O/tune.json
timeout 900 python bench.py --layers --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/bench_layers.txt; head -1 $O/bench.json | cut -c90-240
grep -vE "^tune|^wino|amdgpu|socket" $O/bench_layers.txt | cut -c1-130 | awk 'NR%3==1' | head -30
