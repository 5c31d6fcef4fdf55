#!/bin/bash
# Round 2, session 6: stall attribution of the bf16x3 tiles by ablation (diagnostics build: skip DMA after the first chunk /
# skip barriers; wrong results by design, timing only).
O=gpurun_out/r2s6; mkdir -p $O
export TMPDIR=/tmp
(make -C yolact_amd/csrc clean > /dev/null; make -C yolact_amd/csrc -j16 DIAG=1 2>&1 | grep -E "error|Error") 
for planes in 1 0; do
PROBE_PLANES=$planes timeout 300 python tools/conv_probe.py --shapes 0,1,5 --tiles 5,37,33,35 --reps 4 --ablate 0,1,4,5 2>&1 | grep -v amdgpu > $O/ablate_planes$planes.txt
done
paste $O/ablate_planes1.txt $O/ablate_planes0.txt | cut -c1-190
