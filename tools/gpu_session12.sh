#!/bin/bash
O=gpurun_out/s12; mkdir -p $O
export TMPDIR=/tmp YOLACT_AMD_TUNE_CACHE=$PWD/gpurun_out/s12/tune.json
python bench.py --no-cpu-baseline --steps 5 --warmup 1 > /dev/null 2>&1
for m in B A; do
YOLACT_AMD_HEAD0_STREAM=$m timeout 600 python bench.py --no-cpu-baseline > $O/bench_$m.json 2> $O/err_$m.txt; echo "head0 on $m:"; cut -c100-230 $O/bench_$m.json
done
python tools/detect_probe.py 2>&1 | grep "ablate=0"
timeout 600 python -m pytest tests/test_gpu_path.py -m gpu -x -q 2>&1 | tail -1
