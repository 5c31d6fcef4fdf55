#!/bin/bash
# hipGraph replay: correctness test, then batch-1 / batch-8 throughput with and without the graph.
O=gpurun_out/graph; mkdir -p $O
export TMPDIR=/tmp YOLACT_AMD_TUNE_CACHE=$PWD/$O/tune.json
true
for B in 1 2 8; do
  for G in 0 1; do
    echo "batch $B graph $G: $(YOLACT_AMD_GRAPH=$G timeout 300 python bench.py --batch $B --steps 50 --no-cpu-baseline 2>$O/err_${B}_$G.txt | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "img/s", d["ms_per_step"], "ms")')"
  done
done
tail -3 $O/err_1_1.txt
