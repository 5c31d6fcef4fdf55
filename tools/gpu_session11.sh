#!/bin/bash
O=gpurun_out/s11; mkdir -p $O
export TMPDIR=/tmp YOLACT_AMD_TUNE_CACHE=$PWD/gpurun_out/s11/tune.json
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench2.json 2> $O/err2.txt; cut -c1-330 $O/bench2.json
YOLACT_AMD_STREAMS=1 timeout 600 python bench.py --no-cpu-baseline > $O/bench1.json 2> $O/err1.txt; cut -c1-330 $O/bench1.json
