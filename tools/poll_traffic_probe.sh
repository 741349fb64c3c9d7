#!/bin/bash
# What drives FETCH_SIZE?  (1) GOP 30 vs all-intra (no reference reads at all), (2) optional library variants given as
# arguments (paths), e.g. builds with a longer s_sleep in the poll loops.  Run on the GPU box.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
probe() {   # name, gop
  rm -rf $R/gpurun_out/fz_$1
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/fz_$1 --output-format csv -- python3 $R/tools/clip_debug.py 300 1920 1080 $2 > $R/gpurun_out/fz_$1.log 2>&1
  python3 - <<PY
import csv,glob,re
t=0;n=0
for f in glob.glob("$R/gpurun_out/fz_$1/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "h264e_mb" in r["Kernel_Name"]: t+=float(r["Counter_Value"]); n+=1
print("$1 (gop $2): FETCH %.2f GB over %d launches = %.2f KB per useful macroblock (2 passes x 300 frames)" % (t*1024/1e9, n, t/(2*300*8160)))
PY
}
probe gop30 30
probe intra 1
for lib in "$@"; do export H264E_LIB=$lib; probe $(basename $lib .so) 30; done
