#!/usr/bin/env python3
"""How much of a single-slice constant-QP stream has to be encoded again when it is cut into GOP shards that start from a
speculated mv_clusters state?  All shards run on device 0, one after the other: this measures the
re-encode RATE (a property of the clip and the algorithm), not multi-GPU time.
   shard_probe.py [frames] [w h] [gop] [nshards ...]"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import _pkg  # noqa: E402

P = _pkg()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
gop = int(sys.argv[4]) if len(sys.argv) > 4 else 30
counts = [int(v) for v in sys.argv[5:]] or [2, 4, 8]
for ns in counts:
    for slices in (0, 8):
        shards, t_first = [], []
        for a, b in P.shard_ranges(n, gop, ns):
            s = P.StreamShard(w, h, a, b, gop, 26, slices=slices)
            s.enc.generate_synth(0, b - a, t0=a, seed=1)
            t0 = time.time()
            s.first_pass()
            t_first.append(time.time() - t0)
            shards.append(s)
        state, t_settle = (0, 0), []
        for s in shards:
            t0 = time.time()
            state = s.settle(state)
            t_settle.append(time.time() - t0)
        md5 = hashlib.md5(b"".join(s.bytes() for s in shards)).hexdigest()
        # the time N GPUs would need: all first passes at once, then the settles one after the other down the chain
        est = max(t_first) + sum(t_settle)
        print("%dx%d %d frames gop %d slices %d, %d shards: frames encoded again per shard %s (%.0f %% of the stream); first pass %.3f s max, settle chain %.3f s, "
              "estimated N-GPU time %.3f s vs one encoder %.3f s; md5 %s" %
              (w, h, n, gop, max(slices, 1), len(shards), [s.reencoded for s in shards], 100.0 * sum(s.reencoded for s in shards) / n, max(t_first), sum(t_settle),
               est, sum(t_first) if len(shards) == 1 else 0.0, md5))
        for s in shards:
            s.close()
