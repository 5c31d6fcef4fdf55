#!/bin/bash
# The other BASELINE configs on the current engine (1 GPU, forward + Detect), one bench line each.
O=gpurun_out/configs; mkdir -p $O
export TMPDIR=/tmp
run() { timeout 400 python bench.py --no-cpu-baseline --steps 20 "$@" 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["config"]["workload"][:60], "|", d["value"], "img/s", d["ms_per_step"], "ms | all_conv", r["all_conv"]["tflops"], "TF alg", r["all_conv"]["frac"], "| engine exec", r["engine"]["executed_tflops"])'; }
run --config yolact_base_config --batch 16 | tee $O/base.txt
run --config yolact_im700_config --size 700 --batch 8 | tee $O/im700.txt
run --config yolact_darknet53_config --batch 8 | tee $O/darknet.txt
run --config yolact_plus_resnet50_config --batch 8 | tee $O/plus.txt
run --with-postprocess | tee $O/post.txt
