#!/bin/bash
export TMPDIR=/tmp
for m in 1 2; do
YOLACT_AMD_BATCH_SPLIT=$m timeout 600 python bench.py --no-cpu-baseline > /tmp/b$m.json 2> /tmp/e$m.txt; echo "split $m:"; cut -c100-230 /tmp/b$m.json; tail -2 /tmp/e$m.txt | grep -v amdgpu
done
