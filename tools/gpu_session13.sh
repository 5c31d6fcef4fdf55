#!/bin/bash
O=gpurun_out/s13; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python bench.py --no-cpu-baseline > $O/r50.json 2> $O/r50.err; cut -c100-240 $O/r50.json
timeout 300 python bench.py --no-cpu-baseline --with-postprocess --steps 10 > $O/r50_pp.json 2> $O/r50_pp.err; cut -c100-240 $O/r50_pp.json
timeout 400 python bench.py --no-cpu-baseline --config yolact_base_config --batch 16 --steps 10 --warmup 2 > $O/r101_b16.json 2> $O/r101.err; cut -c1-60,100-240 $O/r101_b16.json; tail -2 $O/r101.err
timeout 400 python bench.py --no-cpu-baseline --config yolact_plus_resnet50_config --steps 10 --warmup 2 > $O/plus.json 2> $O/plus.err; cut -c100-240 $O/plus.json; tail -2 $O/plus.err
timeout 400 python bench.py --no-cpu-baseline --config yolact_im700_config --size 700 --steps 10 --warmup 2 > $O/im700.json 2> $O/im700.err; cut -c100-240 $O/im700.json; tail -2 $O/im700.err
timeout 400 python bench.py --no-cpu-baseline --config yolact_darknet53_config --steps 10 --warmup 2 > $O/dark.json 2> $O/dark.err; cut -c100-240 $O/dark.json; tail -2 $O/dark.err
