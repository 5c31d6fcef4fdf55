#!/bin/bash
O=gpurun_out/s10; mkdir -p $O
export TMPDIR=/tmp YOLACT_AMD_TUNE_CACHE=$PWD/gpurun_out/s10/tune.json
timeout 900 python -m pytest tests/test_gpu_path.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof -- bash -c "cd $R && python bench.py --steps 10 --warmup 2 --no-cpu-baseline" > $R/$O/prof.log 2>&1)
grep -h "topk\|softmax" $O/prof/*/*kernel_stats.csv | cut -c1-60,150-260
tail -1 $O/prof.log | cut -c1-400
