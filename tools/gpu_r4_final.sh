#!/bin/bash
# round-4 closing visit: the whole -m gpu suite, then the configurations table (profiles/r04_configs.txt) and the phase profiles
TAG=${1:-r4_final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?
echo "tests rc=$rc" | tee -a $OUT/tests.log; tail -4 $OUT/tests.log
[ $rc -ne 0 ] && exit $rc
export H264E_QUIET=1
for cfg in "600 1920 1080 30 26 0 0" "600 1920 1080 30 26 8 0" "600 1920 1080 30 26 16 0" "600 1920 1080 1 26 0 0" "240 3840 2160 30 26 0 0" "240 3840 2160 30 26 8 0" "60 7680 4320 30 26 0 0" "60 7680 4320 30 26 2 0" "60 1920 1080 30 26 0 4000" "60 1920 1080 30 26 8 4000" "20 7680 4320 30 26 2 60000" "3000 352 288 30 26 0 0"; do
  timeout -k 10 200 python tools/clip_debug.py $cfg 2>/dev/null | tail -1 >> $OUT/configs.txt || exit 1
done
for b in 2 4; do timeout -k 10 300 python tools/multi_clip_probe.py $b 600 >> $OUT/configs.txt 2>&1 || exit 1; done
timeout -k 10 300 python tools/multi_clip_probe.py 4 600 1920 1080 30 26 8 >> $OUT/configs.txt 2>&1 || exit 1
timeout -k 10 120 python tools/single_frame_latency.py >> $OUT/configs.txt 2>&1 || exit 1
cat $OUT/configs.txt
H264E_WAVES=2 timeout -k 10 300 python tools/phase_profile.py 600 1920 1080 30 > $OUT/phase600.txt 2>&1 || exit 1
H264E_WAVES=2 H264E_RING=2 timeout -k 10 300 python tools/phase_profile.py 30 1920 1080 30 > $OUT/phase_lone.txt 2>&1 || exit 1
H264E_WAVES=3 H264E_RING=2 timeout -k 10 300 python tools/phase_profile.py 30 1920 1080 30 > $OUT/phase_lone_four_waves.txt 2>&1 || exit 1
echo done
