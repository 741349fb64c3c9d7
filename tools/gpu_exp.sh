#!/bin/bash
# scratch visit: the default bench line of the final build
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-exp}; mkdir -p $OUT; cd $R
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1 || { tail -5 $OUT/smoke.log; exit 1; }
tail -1 $OUT/smoke.log
timeout -k 10 900 python bench.py > $OUT/bench.log 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
tail -1 $OUT/bench.log | cut -c1-300
