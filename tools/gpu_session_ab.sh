#!/bin/bash
# A/B of an engine switch with a shared tune cache: bench value per setting, twice, interleaved.
O=gpurun_out/ab; mkdir -p $O
export TMPDIR=/tmp YOLACT_AMD_TUNE_CACHE=$PWD/$O/tune.json
VAR=$1; shift
for rep in 1 2; do
  for V in "$@"; do
    echo "$VAR=$V: $(env $VAR=$V timeout 300 python bench.py --steps 40 --no-cpu-baseline 2>$O/err_$V.txt | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], "img/s", d["ms_per_step"], "ms")')"
  done
done
