#!/bin/bash
# scratch experiment visit: early "all but me" abort -- parity, rates, rate-control timelines
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-exp}; mkdir -p $OUT; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_failures.py "tests/test_gpu_golden_big.py" -m gpu -x -q -k "not 4k_1200" > $OUT/tests.log 2>&1; rc=$?
echo "tests rc=$rc" | tee -a $OUT/tests.log; tail -5 $OUT/tests.log
[ $rc -ne 0 ] && exit $rc
export H264E_QUIET=1
for cfg in "600 1920 1080 30 26 0 0" "600 1920 1080 30 26 8 0" "240 3840 2160 30 26 0 0" "60 7680 4320 30 26 0 0" "60 1920 1080 30 26 0 4000" "60 1920 1080 30 26 8 4000" "20 7680 4320 30 26 2 60000" "3000 352 288 30 26 0 0"; do
  timeout -k 10 200 python tools/clip_debug.py $cfg 2>/dev/null | tail -1 >> $OUT/configs.txt || exit 1
done
timeout -k 10 120 python tools/single_frame_latency.py >> $OUT/configs.txt 2>&1 || exit 1
cat $OUT/configs.txt
unset H264E_QUIET
for cfg in "60 1920 1080 30 26 0 4000" "20 7680 4320 30 26 2 60000"; do
  echo "== $cfg" >> $OUT/timeline.txt
  H264E_DEBUG=1 timeout -k 10 200 python tools/clip_debug.py $cfg 2>&1 | grep "clip launch\|frames:" | tail -32 >> $OUT/timeline.txt || exit 1
done
