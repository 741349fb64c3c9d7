#!/bin/bash
# per-launch timeline of the 600-frame bench stream (H264E_DEBUG) for each kernel variant, + an event-free stream (GOP 600, 8 slices)
TAG=${1:-r3_timeline}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for wv in ${WAVES_LIST:-0 2 4}; do
  H264E_WAVES=$wv timeout -k 10 200 python tools/clip_debug.py 600 1920 1080 30 26 0 0 > $OUT/timeline_w$wv.txt 2>&1 || exit 1
done
for lb in 20 40 80; do
  H264E_LAUNCH_BASE=$lb timeout -k 10 200 python tools/clip_debug.py 600 1920 1080 30 26 0 0 2>/dev/null | tail -1 >> $OUT/launch_base.txt || exit 1
done
cat $OUT/launch_base.txt
tail -30 $OUT/timeline_w0.txt
