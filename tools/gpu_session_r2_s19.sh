#!/bin/bash
# Round 2, session 19: depth-2 pipelining of the per-step host read in bench.py: A/B against the blocking read, same box.
O=gpurun_out/r2s19; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-pipeline > $O/bench_block_$i.json 2> $O/err.txt; head -1 $O/bench_block_$i.json | cut -c90-200
timeout 300 python bench.py --no-cpu-baseline --no-secondary > $O/bench_pipe_$i.json 2>> $O/err.txt; head -1 $O/bench_pipe_$i.json | cut -c90-200
done
timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 > $O/bench_driver_like.json 2>> $O/err.txt; head -1 $O/bench_driver_like.json | cut -c90-200
grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" $O/err.txt | tail -5
