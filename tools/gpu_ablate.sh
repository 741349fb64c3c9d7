#!/bin/bash
# What-if timings: the product against builds that leave one piece of work out (NOT bit-exact; make -C h264-lab_amd/csrc ablate).
# Says how much the lone-frame latency and the event-free stream rate depend on each piece.  Logs under gpurun_out/$1/.
TAG=${1:-ablate}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export H264E_QUIET=1
for n in 0 1 2 3 4; do
  if [ $n = 0 ]; then unset H264E_LIB; name="product"; else export H264E_LIB=$R/h264-lab_amd/lib/libh264e_mi355x_ablate$n.so; name="ablate $n"; fi
  echo "--- $name" >> $OUT/ablate.txt
  timeout -k 10 100 python tools/single_frame_latency.py 2>&1 | tail -1 >> $OUT/ablate.txt || exit 1
  timeout -k 10 200 python tools/clip_debug.py 600 1920 1080 30 26 8 0 2>/dev/null | tail -1 >> $OUT/ablate.txt || exit 1
  timeout -k 10 200 python tools/clip_debug.py 600 1920 1080 30 26 0 0 2>/dev/null | tail -1 >> $OUT/ablate.txt || exit 1
done
cat $OUT/ablate.txt
