#!/bin/bash
# Round 2, session 20: launch-path environment A/B (same box, alternating): HIP_FORCE_DEV_KERNARG, GPU_MAX_HW_QUEUES.
O=gpurun_out/r2s20; mkdir -p $O
export TMPDIR=/tmp
run() { timeout 300 env "$@" python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do
echo "default            : $(run A=1)"
echo "DEV_KERNARG=0      : $(run HIP_FORCE_DEV_KERNARG=0)"
echo "DEV_KERNARG=1      : $(run HIP_FORCE_DEV_KERNARG=1)"
done
echo "HW_QUEUES=4        : $(run GPU_MAX_HW_QUEUES=4)"
echo "HW_QUEUES=16       : $(run GPU_MAX_HW_QUEUES=16)"
echo "batch1 default     : $(timeout 300 python bench.py --batch 1 --steps 50 --no-cpu-baseline --no-secondary 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
echo "batch1 KERNARG=1   : $(HIP_FORCE_DEV_KERNARG=1 timeout 300 python bench.py --batch 1 --steps 50 --no-cpu-baseline --no-secondary 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
echo "batch1 KERNARG=0   : $(HIP_FORCE_DEV_KERNARG=0 timeout 300 python bench.py --batch 1 --steps 50 --no-cpu-baseline --no-secondary 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
