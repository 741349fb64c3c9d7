#!/bin/bash
# H264E_XCD_BANDS A/B: speed (clip_debug) and FETCH_SIZE / WRITE_SIZE per launch (rocprofv3 --pmc, separate passes)
TAG=${1:-r3_bands}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export H264E_QUIET=1
for b in 0 8; do
  echo "--- H264E_XCD_BANDS=$b" >> $OUT/speed.txt
  for cfg in "600 1920 1080 30 26 0 0" "600 1920 1080 30 26 8 0" "240 3840 2160 30 26 0 0" "3000 352 288 30 26 0 0"; do
    H264E_XCD_BANDS=$b timeout -k 10 200 python tools/clip_debug.py $cfg 2>/dev/null | tail -1 >> $OUT/speed.txt || exit 1
  done
done
cat $OUT/speed.txt
cd /tmp; export TMPDIR=/tmp
for b in 0 8; do
  for c in FETCH_SIZE WRITE_SIZE; do
    H264E_XCD_BANDS=$b rocprofv3 --pmc $c -d $OUT/pmc_${c}_$b --output-format csv -- python3 $R/tools/clip_debug.py 600 1920 1080 30 26 0 0 > $OUT/pmc_${c}_$b.log 2>&1 || exit 1
  done
done
python3 - <<PY
import csv, glob
for b in (0, 8):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        tot = n = 0
        for f in glob.glob("$OUT/pmc_%s_%d/*/*_counter_collection.csv" % (c, b)):
            for r in csv.DictReader(open(f)):
                if "h264e_mb_kernel" in r["Kernel_Name"]:
                    tot += float(r["Counter_Value"]); n += 1
        print("H264E_XCD_BANDS=%d %s: %d launches, %.1f MB per launch, %.2f GB per 2 passes" % (b, c, n, tot/max(n,1)/1024, tot/1048576))
PY
