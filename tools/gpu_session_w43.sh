#!/bin/bash
# F(4x4,3x3) bring-up: kernel tests, then bench with the per-layer direct / F(2x2) / F(4x4) table, then the e2e parity tests.
O=gpurun_out/w43; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "winograd" > $O/pytest_k.log 2>&1; tail -3 $O/pytest_k.log
export YOLACT_AMD_TUNE_CACHE=$PWD/$O/tune.json
timeout 600 python bench.py --layers > $O/bench.json 2> $O/bench_layers.txt; cut -c1-300 $O/bench.json; grep "^wino" $O/bench_layers.txt
unset YOLACT_AMD_TUNE_CACHE
timeout 900 python -m pytest tests/test_gpu_path.py tests/test_map_parity.py -q -x > $O/pytest_p.log 2>&1; tail -5 $O/pytest_p.log
