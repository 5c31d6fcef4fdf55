#!/bin/bash
O=gpurun_out/s9; mkdir -p $O
export TMPDIR=/tmp YOLACT_AMD_TUNE_CACHE=$PWD/$O/tune.json
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python bench.py --layers > $O/bench.json 2> $O/bench_layers.txt
cat $O/bench.json | cut -c1-900
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof -- bash -c "cd $R && python bench.py --steps 10 --warmup 2 --no-cpu-baseline" > $R/$O/prof.log 2>&1)
ls -R $O/prof | head -20
