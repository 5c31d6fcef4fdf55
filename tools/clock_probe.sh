#!/bin/bash
# Sample the shader clock / power while a long conv loop runs: is the 130 TF/s ceiling of the big layers a clock limit?
export TMPDIR=/tmp
(python tools/conv_probe.py --shapes 0 --tiles 5 --reps 2000 > /tmp/probe.txt 2>&1) &
PID=$!
sleep 6
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | tr '\n' ' '; echo
  sleep 0.3
done
wait $PID
grep -v amdgpu /tmp/probe.txt
echo "--- idle:"
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo
# pure MFMA microbench for comparison
(tools/microbench.bin > /tmp/mb.txt 2>&1) &
PID=$!
sleep 1.0
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.2; done
wait $PID; cat /tmp/mb.txt | cut -c1-300
