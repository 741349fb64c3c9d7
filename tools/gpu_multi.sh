#!/bin/bash
TAG=${1:-r3_multi}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_golden_big.py -m gpu -x -q -k "launch_group" > $OUT/tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -5 $OUT/tests.log
[ $rc -ne 0 ] && exit $rc
for b in 1 2 3 4; do
  timeout -k 10 300 python tools/multi_clip_probe.py $b 600 >> $OUT/multi.txt 2>&1 || exit 1
done
PROBE_THREADS=1 timeout -k 10 300 python tools/multi_clip_probe.py 4 600 >> $OUT/multi.txt 2>&1 || exit 1
timeout -k 10 300 python tools/multi_clip_probe.py 4 600 1920 1080 30 26 8 >> $OUT/multi.txt 2>&1 || exit 1
cat $OUT/multi.txt
