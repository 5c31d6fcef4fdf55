#!/bin/bash
# Round 2, GPU session 2: (1) where the overlap went (process group vs streams), (2) bf16x3 tiles: unit parity, per-layer
# tune with the split candidates, end-to-end parity + bench with YOLACT_AMD_SPLIT=1.
O=gpurun_out/r2s2; mkdir -p $O
export TMPDIR=/tmp
( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/split_probe.hip -o /tmp/split_probe.bin && timeout 300 /tmp/split_probe.bin && SPLIT_PROBE_ZEROS=1 timeout 300 /tmp/split_probe.bin ) > $O/split_probe.json 2> $O/split_probe.err; cut -c1-3000 $O/split_probe.json
for pg in none before after; do timeout 300 python tools/overlap_probe.py --pg $pg 2>&1 | grep "^{" ; done > $O/overlap.txt
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/overlap_probe.py --pg before 2>&1 | grep "^{" >> $O/overlap.txt
cat $O/overlap.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -k "bf16x3 or conv_plain" -rA > $O/pytest_x3.log 2>&1; grep -E "bf16x3 err|passed|failed" $O/pytest_x3.log | cut -c1-200 | tail -12
export YOLACT_AMD_SPLIT=1
export YOLACT_AMD_TUNE_CACHE=$PWD/$O/tune_x3.json
timeout 900 python bench.py --layers --no-cpu-baseline > $O/bench_x3.json 2> $O/bench_x3_layers.txt; head -1 $O/bench_x3.json | cut -c1-400; grep -E "^tune|^wino" $O/bench_x3_layers.txt | cut -c1-330 | head -90
timeout 1500 python -m pytest tests/test_gpu_batch_parity.py -m gpu -q --timeout 900 -rA -k "timed_plan" > $O/pytest_parity_x3.log 2>&1; grep -E "head errors|passed|failed|Error" $O/pytest_parity_x3.log | cut -c1-420
