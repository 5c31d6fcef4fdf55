#!/bin/bash
O=gpurun_out/s14; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 400 python bench.py --no-cpu-baseline --config yolact_plus_resnet50_config --steps 10 --warmup 2 > $O/plus.json 2> $O/plus.err; cut -c100-240 $O/plus.json
timeout 300 python bench.py --no-cpu-baseline > $O/r50.json 2> $O/r50.err; cut -c100-240 $O/r50.json
