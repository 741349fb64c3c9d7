#!/bin/bash
# scratch visit: row checkpoints (re-encode from a column of a row) -- parity, rates, timeline
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-exp}; mkdir -p $OUT; cd $R
timeout -k 10 900 python -m pytest tests/test_stages.py tests/test_gpu_parity.py tests/test_gpu_failures.py "tests/test_gpu_golden_big.py" -m gpu -x -q -k "not 4k_1200" > $OUT/tests.log 2>&1; rc=$?
echo "tests rc=$rc" | tee -a $OUT/tests.log; tail -5 $OUT/tests.log
[ $rc -ne 0 ] && exit $rc
export H264E_QUIET=1
for ck in "" 1; do
for cfg in "600 1920 1080 30 26 0 0" "600 1920 1080 30 26 8 0" "240 3840 2160 30 26 0 0" "60 7680 4320 30 26 0 0" "60 1920 1080 30 26 0 4000" "60 1920 1080 30 26 8 4000" "20 7680 4320 30 26 2 60000" "3000 352 288 30 26 0 0" "600 1280 720 30 26 0 0"; do
  echo "NO_CK=$ck $cfg" >> $OUT/configs.txt
  env ${ck:+H264E_NO_ROW_CHECKPOINTS=1} timeout -k 10 200 python tools/clip_debug.py $cfg 2>/dev/null | tail -1 >> $OUT/configs.txt || exit 1
done; done
timeout -k 10 120 python tools/single_frame_latency.py >> $OUT/configs.txt 2>&1 || exit 1
unset H264E_QUIET
H264E_DEBUG=1 timeout -k 10 200 python tools/clip_debug.py 600 1920 1080 30 26 0 0 2>&1 | grep "clip launch" | tail -24 > $OUT/timeline.txt
