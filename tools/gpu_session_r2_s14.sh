#!/bin/bash
# Round 2, session 14: pull_item row — JPEG reconstruction kernels + COCODetection end to end; half-batch overlap probe.
O=gpurun_out/r2s14; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_jpeg.py -m gpu -q --timeout 300 -rA > $O/pytest_jpeg.log 2>&1; tail -8 $O/pytest_jpeg.log | cut -c1-300
timeout 200 python tools/jpeg_probe.py > $O/jpeg_probe.json 2> $O/jpeg_probe.err; cat $O/jpeg_probe.json; tail -2 $O/jpeg_probe.err | cut -c1-200
export YOLACT_AMD_TUNE_CACHE=$PWD/$O/tune_b4.json
timeout 500 python tools/halfbatch_probe.py > $O/halfbatch.json 2> $O/halfbatch.err; cat $O/halfbatch.json; tail -2 $O/halfbatch.err | cut -c1-200
