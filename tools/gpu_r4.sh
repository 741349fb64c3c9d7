#!/bin/bash
# round-4 iteration visit: (FULL=1: the whole -m gpu suite | quick parity subset), rates of the main configurations, lone-frame latency,
# per-phase profiles of the stamps build (rows + finalizer).  Logs under gpurun_out/$1/.
TAG=${1:-r4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
if [ -n "$FULL" ]; then
  timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?
elif [ -z "$NOTESTS" ]; then
  timeout -k 10 600 python -m pytest tests/test_stages.py tests/test_gpu_parity.py "tests/test_gpu_golden_big.py::test_full_length_stream_matches_reference" -m gpu -x -q -k "not 4k_ and not 8k_" > $OUT/tests.log 2>&1; rc=$?
else rc=0; fi
echo "tests rc=$rc" | tee -a $OUT/tests.log
tail -5 $OUT/tests.log
[ $rc -ne 0 ] && exit $rc
export H264E_QUIET=1
for cfg in "600 1920 1080 30 26 0 0" "600 1920 1080 30 26 8 0" "600 1920 1080 1 26 0 0" ${EXTRA_CFGS}; do
  timeout -k 10 200 python tools/clip_debug.py $cfg 2>/dev/null | tail -1 >> $OUT/configs.txt || exit 1
done
timeout -k 10 120 python tools/single_frame_latency.py >> $OUT/configs.txt 2>&1 || exit 1
cat $OUT/configs.txt
if [ -z "$NOSTAMPS" ]; then
  H264E_WAVES=2 timeout -k 10 300 python tools/phase_profile.py 600 1920 1080 30 > $OUT/phase600.txt 2>&1 || exit 1
  H264E_WAVES=2 H264E_RING=2 timeout -k 10 300 python tools/phase_profile.py 30 1920 1080 30 > $OUT/phase_lone.txt 2>&1 || exit 1
  cat $OUT/phase600.txt $OUT/phase_lone.txt
fi
