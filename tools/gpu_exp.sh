#!/bin/bash
# one-off GPU visit: issue priorities (s_setprio by position in the launch) on / off
cd ${GRAFT_REPO_ROOT:-.}
export H264E_QUIET=1
for p in 0 1; do
  echo "--- H264E_PRIO=$p"
  for cfg in "600 1920 1080 30 26 0 0" "600 1920 1080 30 26 8 0" "60 1920 1080 30 26 0 4000" "60 1920 1080 30 26 8 4000" "240 3840 2160 30 26 0 0" "3000 352 288 30 26 0 0"; do H264E_PRIO=$p timeout -k 10 200 python tools/clip_debug.py $cfg 2>/dev/null | tail -1; done
  H264E_PRIO=$p timeout -k 10 120 python tools/single_frame_latency.py
done
unset H264E_QUIET
H264E_PRIO=1 timeout -k 10 200 python tools/clip_debug.py 600 1920 1080 30 26 0 0 2>&1 | tail -26 | cut -c1-200
