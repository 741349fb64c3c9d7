#!/bin/bash
# scratch visit: what the driver runs at round end -- the whole -m gpu suite, smoke, the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-exp}; mkdir -p $OUT; cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1; rc=$?
tail -3 $OUT/tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1 || { tail -5 $OUT/smoke.log; exit 1; }
tail -1 $OUT/smoke.log
timeout -k 10 900 python bench.py > $OUT/bench.log 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
tail -1 $OUT/bench.log | cut -c1-200
