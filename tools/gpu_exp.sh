#!/bin/bash
# scratch visit: determinism soak
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-exp}; mkdir -p $OUT; cd $R
timeout -k 10 600 python tools/soak_determinism.py > $OUT/soak.txt 2>&1; rc=$?
cat $OUT/soak.txt | tail -12; exit $rc
