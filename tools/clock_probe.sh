#!/bin/bash
# sample the GPU clocks while the encoder runs (fast/slow mode investigation)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3 4 5 6; do
  ( for k in $(seq 1 8); do rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk\|socclk" | sed 's/.*GPU\[0\]\t*: *\([a-z]*\) clock level: *\([0-9S]*\): *(\(.*\))/\1=\3/' | tr '\n' ' '; echo; sleep 0.15; done ) > $R/gpurun_out/clk_$i.txt &
  SP=$!
  sleep 0.4
  timeout -k 10 200 python $R/tools/clip_debug.py 600 2>&1 | tail -1
  wait $SP
  sed -n 4,7p $R/gpurun_out/clk_$i.txt
done
