#!/bin/bash
# scratch experiment visit: XCD bands with padded (equally long) queues -- speed everywhere, FETCH_SIZE at 1080p
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-exp}; mkdir -p $OUT; cd $R
export H264E_QUIET=1
for b in 0 8; do
  for cfg in "600 1920 1080 30 26 0 0" "600 1920 1080 30 26 8 0" "600 1280 720 30 26 0 0" "240 3840 2160 30 26 0 0" "240 3840 2160 30 26 8 0" "60 7680 4320 30 26 0 0" "60 7680 4320 30 26 2 0" "3000 352 288 30 26 0 0" "60 1920 1080 30 26 0 4000"; do
    echo "BANDS=$b $cfg" >> $OUT/bands.txt
    H264E_XCD_BANDS=$b timeout -k 10 200 python tools/clip_debug.py $cfg 2>&1 | tail -1 >> $OUT/bands.txt
  done
done
H264E_XCD_BANDS=8 timeout -k 10 120 python tools/single_frame_latency.py >> $OUT/bands.txt 2>&1
cd /tmp; export TMPDIR=/tmp
for b in 8; do
  H264E_XCD_BANDS=$b timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch_$b --output-format csv -- python3 $R/tools/clip_debug.py 600 1920 1080 30 26 0 0 > $OUT/fetch_$b.log 2>&1 || exit 1
  H264E_XCD_BANDS=$b timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/write_$b --output-format csv -- python3 $R/tools/clip_debug.py 600 1920 1080 30 26 0 0 > $OUT/write_$b.log 2>&1 || exit 1
done
python3 - <<PY
import csv,glob,collections
for tag in ("8",):
    for c in ("fetch","write"):
        s=0;n=0
        for f in glob.glob("$OUT/%s_%s/*/*_counter_collection.csv"%(c,tag)):
            for r in csv.DictReader(open(f)):
                if "h264e_mb_kernel" in r["Kernel_Name"]: s+=float(r["Counter_Value"]); n+=1
        print("bands %s %s: %.0f MB per launch over %d launches"%(tag,c,s/1024/max(n,1),n))
PY
