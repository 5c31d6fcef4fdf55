#!/bin/bash
O=gpurun_out/s6; mkdir -p $O
export TMPDIR=/tmp
cd tools
(for args in "0 5" "2 3"; do python conv_trace.py $args; echo; done
export HIP_FORCE_DEV_KERNARG=1
echo "=== HIP_FORCE_DEV_KERNARG=1"
for args in "0 5" "2 3"; do python conv_trace.py $args; echo; done
cd ..; python bench.py --no-cpu-baseline --steps 20 | cut -c1-700
export HIP_FORCE_DEV_KERNARG=0
echo "=== HIP_FORCE_DEV_KERNARG=0"
python bench.py --no-cpu-baseline --steps 20 | cut -c1-700 ) > ../$O/trace.txt 2>&1
