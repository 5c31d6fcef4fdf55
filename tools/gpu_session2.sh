#!/bin/bash
set -x
O=gpurun_out/s2; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 600 python tools/conv_probe.py --ablate 0 --reps 10 > $O/probe.txt 2>&1
timeout 600 python bench.py --layers --no-cpu-baseline > $O/bench.json 2> $O/bench_layers.txt
cat $O/bench.json
