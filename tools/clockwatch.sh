#!/bin/bash
# tools/clockwatch.sh -- sample clocks and power while a kernel benchmark runs (GPU box)
out=gpurun_out/clockwatch.log
: > $out
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (edge|junction|hbm)" >> $out; echo "---" >> $out; sleep 0.5; done ) &
W=$!
F=64 STEPS=60 timeout -k 10 200 python tools/kbench.py c2box 2>&1 | grep -v amdgpu.ids
kill $W 2>/dev/null
wait $W 2>/dev/null
grep -E "sclk|Power" $out | sort | uniq -c | sort -rn | head -30
