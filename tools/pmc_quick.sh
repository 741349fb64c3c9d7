#!/bin/bash
# quick instruction count of the macroblock kernel for one configuration (rocprofv3 --pmc, counters only; run on the GPU box):
#   tools/pmc_quick.sh <tag> [clip_debug.py arguments]      -> gpurun_out/pmc_<tag>.txt
# Two passes of the clip (clip_debug warms up once); instructions are reported per USEFUL macroblock of ONE pass pair.
TAG=${1:-q}; shift
ARGS=${@:-600 1920 1080 30 26 0 0}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcq_$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
H264E_QUIET=1 timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM -d $OUT/p1 --output-format csv -- python3 $R/tools/clip_debug.py $ARGS > $OUT/p1.log 2>&1 || { tail -5 $OUT/p1.log; exit 1; }
python3 - "$ARGS" > $R/gpurun_out/pmc_$TAG.txt <<PY
import csv,glob,collections,sys
a=sys.argv[1].split(); n,w,h=int(a[0]),int(a[1]),int(a[2])
agg=collections.defaultdict(float)
for f in glob.glob('$OUT/p1/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'h264e_mb_kernel' in r['Kernel_Name']: agg[r['Counter_Name']]+=float(r['Counter_Value'])
useful=2.0*n*((w+15)//16)*((h+15)//16)
print("h264e_mb_kernel, clip_debug.py %s, two passes; per useful macroblock: VALU %.0f  SALU %.0f  LDS %.0f  (waves %d)" % (sys.argv[1], agg['SQ_INSTS_VALU']/useful, agg['SQ_INSTS_SALU']/useful, agg['SQ_INSTS_LDS']/useful, agg['SQ_WAVES']))
PY
tail -1 $OUT/p1.log; cat $R/gpurun_out/pmc_$TAG.txt
