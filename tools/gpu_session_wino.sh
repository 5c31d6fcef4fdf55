#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline --layers > /tmp/b.json 2>/tmp/e.txt; cut -c100-230 /tmp/b.json; tail -2 /tmp/e.txt | grep -v amdgpu
python - <<'PY'
import json
d=json.load(open('/tmp/b.json')); r=d['roofline']; print(r['kernel'], r['achieved'], r['frac'], r['traffic'], {k:v for k,v in r['all_conv'].items() if k!='basis'})
for k,v in sorted(r['per_kernel'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:8]: print(k, round(v['ms_per_step'],3), round(v['tflops'],1), v['launches_per_step'])
PY
grep "winograd" /tmp/e.txt | head -40
