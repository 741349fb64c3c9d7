#!/bin/bash
# round-3 iteration visit: stage fixtures + parity on the GPU, the 600-frame bench stream (md5-checked by the golden test), rates, per-phase profile.
# Logs under gpurun_out/$1/.  FULL=1 runs the whole -m gpu suite instead of the quick subset.
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
if [ -n "$FULL" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?
else
  timeout -k 10 600 python -m pytest tests/test_stages.py tests/test_gpu_parity.py "tests/test_gpu_golden_big.py::test_full_length_stream_matches_reference" -m gpu -x -q -k "not 4k_240 and not 8k_3" > $OUT/tests.log 2>&1; rc=$?
fi
echo "tests rc=$rc" | tee -a $OUT/tests.log
tail -5 $OUT/tests.log
[ $rc -ne 0 ] && exit $rc
export H264E_QUIET=1
for wv in ${WAVES_LIST:-2 1}; do
  export H264E_WAVES=$wv
  echo "--- H264E_WAVES=$wv" >> $OUT/configs.txt
  for cfg in "600 1920 1080 30 26 0 0" "600 1920 1080 30 26 8 0" "600 1920 1080 1 26 0 0" ${EXTRA_CFGS}; do
    timeout -k 10 200 python tools/clip_debug.py $cfg 2>/dev/null | tail -1 >> $OUT/configs.txt || exit 1
  done
  timeout -k 10 120 python tools/single_frame_latency.py >> $OUT/configs.txt 2>&1 || exit 1
done
cat $OUT/configs.txt
H264E_WAVES=1 timeout -k 10 300 python tools/phase_profile.py 600 1920 1080 30 > $OUT/phase600.txt 2>&1 || exit 1
cat $OUT/phase600.txt
