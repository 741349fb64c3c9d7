#!/bin/bash
# scratch visit: stability soak
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-exp}; mkdir -p $OUT; cd $R
timeout -k 10 1000 python tools/stress_r04.py > $OUT/stress.txt 2>&1; rc=$?
tail -14 $OUT/stress.txt; exit $rc
