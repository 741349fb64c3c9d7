#!/bin/bash
# round-3 measurement visit: bench line, extra configurations, multi-stream aggregate, per-phase profiles.  Logs under gpurun_out/$1/.
TAG=${1:-r3m}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 400 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -c 600 $OUT/bench.json
export H264E_QUIET=1
for cfg in "600 1920 1080 30 26 0 0" "600 1920 1080 30 26 8 0" "600 1920 1080 30 26 16 0" "600 1920 1080 1 26 0 0" "240 3840 2160 30 26 0 0" "240 3840 2160 30 26 8 0" "60 7680 4320 30 26 0 0" "60 7680 4320 30 26 2 0" "60 1920 1080 30 26 0 4000" "60 1920 1080 30 26 8 4000" "20 7680 4320 30 26 2 60000" "3000 352 288 30 26 0 0"; do
  timeout -k 10 200 python tools/clip_debug.py $cfg 2>/dev/null | tail -1 >> $OUT/configs.txt || exit 1
done
cat $OUT/configs.txt
for b in 2 4; do timeout -k 10 300 python tools/multi_clip_probe.py $b 600 >> $OUT/multi.txt 2>&1 || exit 1; done
PROBE_STAGGER=1 timeout -k 10 300 python tools/multi_clip_probe.py 4 600 >> $OUT/multi.txt 2>&1 || exit 1
timeout -k 10 300 python tools/multi_clip_probe.py 4 600 1920 1080 30 26 8 >> $OUT/multi.txt 2>&1 || exit 1
cat $OUT/multi.txt
timeout -k 10 120 python tools/single_frame_latency.py > $OUT/lone.txt 2>&1; cat $OUT/lone.txt
bash tools/gpu_prof.sh $TAG > /dev/null 2>&1
cat $OUT/phase600_w2.txt $OUT/phase_lone_w2.txt
