#!/bin/bash
# SQ instruction / wait counters of the macroblock kernel (rocprofv3 --pmc, one group per run; no tracing in the same
# run); run on the GPU box.  Summary: gpurun_out/pmc/summary.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM -d $OUT/p1 --output-format csv -- python3 $R/tools/clip_debug.py 120 > $OUT/p1.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/p2 --output-format csv -- python3 $R/tools/clip_debug.py 120 > $OUT/p2.log 2>&1
python3 - > $OUT/summary.txt <<PY
import csv,glob,collections
agg=collections.defaultdict(float)
for p in ['p1','p2']:
    for f in glob.glob('$OUT/'+p+'/*/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'h264e_mb_kernel' in r['Kernel_Name']: agg[r['Counter_Name']]+=float(r['Counter_Value'])
print("h264e_mb_kernel, 1920x1080 x 120 frames GOP 30 QP 26, two passes of the clip (tools/pmc_insts.sh); sums over all launches")
for k,v in sorted(agg.items()): print("%-22s %.4g" % (k, v))
wc=agg.get('SQ_WAVE_CYCLES',0)
if wc:
    print("wave cycles: parked on s_waitcnt/sleep/barrier (SQ_WAIT_ANY) %.1f %%, issue stalls (SQ_WAIT_INST_ANY) %.1f %%, issuing (SQ_ACTIVE_INST_ANY) %.1f %%" %
          (100*agg['SQ_WAIT_ANY']/wc, 100*agg['SQ_WAIT_INST_ANY']/wc, 100*agg['SQ_ACTIVE_INST_ANY']/wc))
PY
cat $OUT/summary.txt
