#!/bin/bash
O=gpurun_out/s4; mkdir -p $O
export TMPDIR=/tmp
cd tools
for args in "2 3" "2 3 1" "2 3 5" "2 7" "0 3" "3 7" "3 3" "1 3"; do python conv_trace.py $args; echo; done > ../$O/trace.txt 2>&1
