#!/bin/bash
# rocprofv3 passes over the bench command (run on the GPU box): kernel trace + stats, then FETCH_SIZE and WRITE_SIZE
# in separate --pmc runs (MI355X_MICROARCH.md: the TCC block cannot hold both).  Output under gpurun_out/prof_$1/.
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch --output-format csv -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write --output-format csv -- $CMD > $OUT/write.log 2>&1
grep -h '"metric"' $OUT/trace.log $OUT/fetch.log $OUT/write.log || true
find $OUT -name "*.csv" | head -20
