#!/bin/bash
# one-off GPU visit (round 4): the full-length 4K goldens and the shard re-encode probe at 4K x 1200
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r4_4k
timeout -k 10 500 python -m pytest "tests/test_gpu_golden_big.py::test_full_length_stream_matches_reference" -m gpu -x -q -k "4k_1200" > gpurun_out/r4_4k/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4_4k/tests.log
H264E_QUIET=1 timeout -k 10 500 python tools/shard_probe.py 1200 3840 2160 30 1 8 > gpurun_out/r4_4k/shard_probe.txt 2>&1; echo "probe rc=$?"; cat gpurun_out/r4_4k/shard_probe.txt
