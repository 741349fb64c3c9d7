#!/bin/bash
# rocprofv3 passes of round 4 over the bench command (run on the GPU box): kernel trace + stats, FETCH_SIZE and WRITE_SIZE in
# separate --pmc runs (MI355X_MICROARCH.md: the TCC block cannot hold both), SQ instruction counters.  Output under gpurun_out/prof_r04/.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r04
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch --output-format csv -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write --output-format csv -- $CMD > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES -d $OUT/sq1 --output-format csv -- $CMD > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/sq2 --output-format csv -- $CMD > $OUT/sq2.log 2>&1
grep -h '"metric"' $OUT/trace.log > $OUT/bench_line_under_trace.json || true
python3 - <<PY
import csv, glob, collections, json
out = "$OUT"
# kernel stats
for f in glob.glob(out + "/trace/*/*_kernel_stats.csv"):
    open(out + "/kernel_stats.csv", "w").write(open(f).read())
agg = collections.defaultdict(float); launches = collections.Counter()
for p in ["fetch", "write", "sq1", "sq2"]:
    for f in glob.glob(out + "/" + p + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "h264e_mb_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); launches[r["Counter_Name"]] += 1
json.dump({"counters": dict(agg), "launches": dict(launches)}, open(out + "/counters.json", "w"), indent=1)
print(json.dumps({"counters": dict(agg), "launches": dict(launches)}))
PY
