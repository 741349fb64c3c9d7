#!/bin/bash
# One GPU-box visit: parity tests, a short bench, the per-phase cycle profile of the stamps build.  Logs under gpurun_out/$1/.
TAG=${1:-chk}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/tests.log
tail -3 $OUT/tests.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 ${BENCH_ARGS} > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json
timeout -k 10 300 python tools/phase_profile.py 600 1920 1080 30 > $OUT/phase600.txt 2>&1; echo "phase rc=$?"
cat $OUT/phase600.txt
