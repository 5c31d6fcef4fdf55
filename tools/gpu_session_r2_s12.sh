#!/bin/bash
# Round 2, session 12: padded class rows (conf_ld) + hoisted segment stores: kernel / path tests, forced re-tune bench with and without padding.
O=gpurun_out/r2s12; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_path.py -m gpu -q --timeout 600 -x > $O/pytest_a.log 2>&1; tail -3 $O/pytest_a.log | cut -c1-300
export YOLACT_AMD_AUTOTUNE=force
export YOLACT_AMD_TUNE_CACHE=$PWD/$O/tune.json
timeout 900 python bench.py --layers --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/bench_layers.txt; echo "stdout lines: $(wc -l < $O/bench.json)"; head -1 $O/bench.json | cut -c90-240
export YOLACT_AMD_TUNE_CACHE=$PWD/$O/tune_nopad.json
YOLACT_AMD_PAD_CONF=0 timeout 900 python bench.py --layers --no-cpu-baseline --no-secondary > $O/bench_nopad.json 2> $O/bench_layers_nopad.txt; head -1 $O/bench_nopad.json | cut -c90-240
grep -E "head|detect" $O/bench_layers.txt | cut -c1-150 | head -12
echo; grep -E "head|detect" $O/bench_layers_nopad.txt | cut -c1-150 | head -12
