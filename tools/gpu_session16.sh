#!/bin/bash
O=gpurun_out/s16; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python tools/conv_probe.py --ablate 0 --reps 10 --tiles 5,1,16,17,18 --shapes 0,1,2,4,5 > $O/probe.txt 2>&1
grep -v amdgpu $O/probe.txt
