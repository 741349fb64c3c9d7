#!/bin/bash
# fast/slow mode investigation: busy-CU counters (and TLB misses in a second group) of the macroblock kernel over several
# processes; run on the GPU box
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_mode
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
  if [ $((i % 2)) = 1 ]; then C="SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; else C="TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_MISS"; fi
  timeout -k 10 90 rocprofv3 --pmc $C -d $OUT/r$i --output-format csv -- python3 $R/tools/clip_debug.py 600 > $OUT/r$i.log 2>&1 || { echo "run $i failed"; tail -2 $OUT/r$i.log; break; }
  grep "^time" $OUT/r$i.log | cut -c1-50
  python3 - <<PY
import csv,glob,collections
for f in glob.glob('$OUT/r$i/*/*_counter_collection.csv'):
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'h264e_mb' in r['Kernel_Name']: agg[r['Counter_Name']]+=float(r['Counter_Value'])
    print({k:("%.4g"%v) for k,v in sorted(agg.items())})
PY
done
