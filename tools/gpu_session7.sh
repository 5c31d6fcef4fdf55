#!/bin/bash
O=gpurun_out/s7; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cd tools
for args in "2 3" "0 5" "7 3"; do python conv_trace.py $args; echo; done > ../$O/trace.txt 2>&1
cd ..
timeout 600 python bench.py --layers --no-cpu-baseline > $O/bench.json 2> $O/bench_layers.txt
cat $O/bench.json
