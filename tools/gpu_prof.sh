#!/bin/bash
# per-phase cycle stamps of the two-wave pipeline: full stream (chip full) and one frame in flight (H264E_RING=2: pure latency)
TAG=${1:-r3_prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
H264E_WAVES=2 timeout -k 10 300 python tools/phase_profile.py 600 1920 1080 30 > $OUT/phase600_w2.txt 2>&1 || exit 1
H264E_WAVES=2 H264E_RING=2 timeout -k 10 300 python tools/phase_profile.py 30 1920 1080 30 > $OUT/phase_lone_w2.txt 2>&1 || exit 1
H264E_WAVES=1 H264E_RING=2 timeout -k 10 300 python tools/phase_profile.py 30 1920 1080 30 > $OUT/phase_lone_w1.txt 2>&1 || exit 1
cat $OUT/phase600_w2.txt $OUT/phase_lone_w2.txt $OUT/phase_lone_w1.txt
