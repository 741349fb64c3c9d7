#!/bin/bash
# instruction-cache and wait counters of the macroblock kernel (rocprofv3 --pmc, one group per run); run on the GPU box
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_ic
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $OUT/a --output-format csv -- python3 $R/tools/clip_debug.py 600 > $OUT/a.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_BUSY_CYCLES -d $OUT/b --output-format csv -- python3 $R/tools/clip_debug.py 600 > $OUT/b.log 2>&1
python3 - <<PY
import csv,glob,collections
for p in ['a','b']:
    for f in glob.glob('$OUT/'+p+'/*/*_counter_collection.csv'):
        agg=collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if 'h264e_mb' in r['Kernel_Name']: agg[r['Counter_Name']]+=float(r['Counter_Value'])
        for k,v in sorted(agg.items()): print(k,v)
PY
