#!/bin/bash
# scratch experiment visit: multi-slice rates with the walk off the verdict chain + parity of the multi-slice goldens
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-exp}; mkdir -p $OUT; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_failures.py "tests/test_gpu_golden_big.py" -m gpu -x -q -k "not 4k_1200" > $OUT/tests.log 2>&1; rc=$?
echo "tests rc=$rc" | tee -a $OUT/tests.log; tail -5 $OUT/tests.log
[ $rc -ne 0 ] && exit $rc
export H264E_QUIET=1
for cfg in "600 1920 1080 30 26 8 0" "600 1920 1080 30 26 2 0" "600 1920 1080 30 26 16 0" "600 1920 1080 30 26 0 0" "240 3840 2160 30 26 8 0" "60 7680 4320 30 26 2 0" "60 1920 1080 30 26 8 4000" "20 7680 4320 30 26 2 60000"; do
  H264E_FZ_WAIT_ALL=$fz timeout -k 10 200 python tools/clip_debug.py $cfg 2>/dev/null | tail -1 >> $OUT/configs.txt || exit 1
done
timeout -k 10 300 python tools/multi_clip_probe.py 4 600 1920 1080 30 26 8 >> $OUT/configs.txt 2>&1 || exit 1
cat $OUT/configs.txt
