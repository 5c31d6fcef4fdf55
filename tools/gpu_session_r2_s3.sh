#!/bin/bash
# Round 2, GPU session 3: bf16x3 with deep-pipeline / big tiles; HW-queue fix in bench.
O=gpurun_out/r2s3; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -k "bf16x3" -rA > $O/pytest_x3.log 2>&1; grep -E "bf16x3 err|passed|failed" $O/pytest_x3.log | grep -v print | cut -c1-200 | tail -12
timeout 600 python bench.py --no-cpu-baseline > $O/bench_fp32.json 2> $O/bench_fp32.err; head -1 $O/bench_fp32.json | cut -c1-330
export YOLACT_AMD_SPLIT=1
export YOLACT_AMD_TUNE_CACHE=$PWD/$O/tune_x3.json
timeout 900 python bench.py --layers --no-cpu-baseline > $O/bench_x3.json 2> $O/bench_x3_layers.txt; head -1 $O/bench_x3.json | cut -c1-400
grep -E "^tune|^wino" $O/bench_x3_layers.txt | python -c "
import sys,re,ast
for l in sys.stdin:
    if l.startswith('tune'):
        name=l.split()[1]; best=l.split()[3]; d=ast.literal_eval(l[l.index('{'):])
        top=sorted(d.items(), key=lambda kv:kv[1])[:6]
        print('%-18s %-12s %s' % (name, best, ' '.join('%s=%.4f'%kv for kv in top)))
    else: print(l.rstrip()[:150])
"
