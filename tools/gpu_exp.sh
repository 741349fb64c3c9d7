#!/bin/bash
# scratch visit: instruction counts per macroblock of configurations without discards (8 slices, all-intra) beside the headline
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bash tools/pmc_quick.sh headline 600 1920 1080 30 26 0 0 || exit 1
bash tools/pmc_quick.sh slices8 600 1920 1080 30 26 8 0 || exit 1
bash tools/pmc_quick.sh intra 600 1920 1080 1 26 0 0 || exit 1
bash tools/pmc_quick.sh rc 60 1920 1080 30 26 0 4000 || exit 1
