#!/bin/bash
# scratch experiment visit: slice counts and concurrent streams under forced kernel variants
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-exp}; mkdir -p $OUT; cd $R
export H264E_QUIET=1
for w in 2 4; do
  for cfg in "600 1920 1080 30 26 2 0" "600 1920 1080 30 26 4 0" "240 3840 2160 30 26 2 0" "240 3840 2160 30 26 4 0" "60 7680 4320 30 26 2 0" "600 1280 720 30 26 8 0"; do
    echo "H264E_WAVES=$w $cfg" >> $OUT/var.txt
    H264E_WAVES=$w timeout -k 10 200 python tools/clip_debug.py $cfg 2>/dev/null | tail -1 >> $OUT/var.txt || exit 1
  done
  for b in 2 4; do echo "H264E_WAVES=$w streams $b" >> $OUT/var.txt; H264E_WAVES=$w timeout -k 10 300 python tools/multi_clip_probe.py $b 600 2>&1 | tail -1 >> $OUT/var.txt || exit 1; done
done
cat $OUT/var.txt
