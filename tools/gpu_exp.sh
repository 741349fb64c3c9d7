#!/bin/bash
# scratch visit: poll interval A/B (s_sleep between two polls of a device-memory counter)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-exp}; mkdir -p $OUT; cd $R
export H264E_QUIET=1
for rep in 1 2; do for n in 8 2 1 16; do
  lib=$R/h264-lab_amd/lib/libh264e_sleep$n.so; [ $n = 8 ] && lib=$R/h264-lab_amd/lib/libh264e_mi355x.so
  for cfg in "600 1920 1080 30 26 0 0" "600 1920 1080 30 26 8 0" "60 1920 1080 30 26 0 4000"; do
    echo "SLEEP=$n $cfg" >> $OUT/sleep.txt
    H264E_LIB=$lib timeout -k 10 200 python tools/clip_debug.py $cfg 2>&1 | tail -1 >> $OUT/sleep.txt
  done
  echo "SLEEP=$n lone" >> $OUT/sleep.txt
  H264E_LIB=$lib timeout -k 10 120 python tools/single_frame_latency.py 2>&1 | tail -1 >> $OUT/sleep.txt
done; done
