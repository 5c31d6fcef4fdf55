#!/bin/bash
set -x
O=gpurun_out/s3; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/conv_probe.py --ablate 0,8 --reps 10 --tiles 3,9,5,6,7,8 > $O/probe.txt 2>&1
timeout 600 python bench.py --layers --no-cpu-baseline > $O/bench.json 2> $O/bench_layers.txt
cat $O/bench.json
