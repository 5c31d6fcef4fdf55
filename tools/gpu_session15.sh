#!/bin/bash
O=gpurun_out/s15; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "dcn or plus" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 400 python bench.py --no-cpu-baseline --config yolact_plus_resnet50_config --steps 10 --warmup 2 > $O/plus.json 2> $O/plus.err; cut -c100-240 $O/plus.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/s15/plus.json')); r=d['roofline']
print(r['all_conv'])
for k,v in sorted(r['per_kernel'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:5]: print(k, round(v['ms_per_step'],3), round(v['tflops'],1), v['launches_per_step'])
PY
