#!/bin/bash
# Round 2, session 9: split-K for small-map 1x1 convolutions: unit tests, forced re-tune of configs[1], bench.
O=gpurun_out/r2s9; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -k "split_k" > $O/pytest_splitk.log 2>&1; tail -4 $O/pytest_splitk.log | cut -c1-300
export YOLACT_AMD_AUTOTUNE=force
export YOLACT_AMD_TUNE_CACHE=$PWD/$O/tune_sk.json
timeout 900 python bench.py --layers --no-cpu-baseline --no-secondary > $O/bench_sk.json 2> $O/bench_sk_layers.txt; head -1 $O/bench_sk.json | cut -c1-330
grep -E "^tune" $O/bench_sk_layers.txt | python -c "
import sys,ast
for l in sys.stdin:
    name=l.split()[1]; best=l.split()[3]; d=ast.literal_eval(l[l.index('{'):])
    top=sorted(d.items(), key=lambda kv:kv[1])[:5]
    sk=[kv for kv in sorted(d.items(), key=lambda kv:kv[1]) if '/k' in kv[0]][:2]
    print('%-18s %-14s %s | %s' % (name, best, ' '.join('%s=%.4f'%kv for kv in top), ' '.join('%s=%.4f'%kv for kv in sk)))
"
grep -vE "^tune|^wino|amdgpu|socket" $O/bench_sk_layers.txt | cut -c1-130 | sed -n 25,60p
