#!/bin/bash
# round-2 measurement visit: GPU tests, bench line, extra configurations, shard re-encode rates.  Logs under gpurun_out/$1/.
TAG=${1:-r2m}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/tests.log
tail -4 $OUT/tests.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json
export H264E_QUIET=1
for cfg in "600 1920 1080 30 26 0 0" "600 1920 1080 30 26 8 0" "600 1920 1080 30 26 16 0" "600 1920 1080 1 26 0 0" "240 3840 2160 30 26 0 0" "240 3840 2160 30 26 8 0" "60 7680 4320 30 26 0 0" "60 7680 4320 30 26 2 0" "60 1920 1080 30 26 0 4000" "60 1920 1080 30 26 8 4000" "20 7680 4320 30 26 2 60000" "3000 352 288 30 26 0 0"; do
  timeout -k 10 200 python tools/clip_debug.py $cfg 2>/dev/null | tail -1 >> $OUT/configs.txt
done
cat $OUT/configs.txt
timeout -k 10 400 python tools/shard_probe.py 600 1920 1080 30 1 2 4 8 > $OUT/shards.txt 2>&1; cat $OUT/shards.txt
