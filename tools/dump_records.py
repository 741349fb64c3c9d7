#!/usr/bin/env python3
"""Per-macroblock records {mv[0], type, consumed-the-candidates} of every frame of a synthetic clip, for offline studies of the mv_clusters
speculation (tools/clusters_sim.py):  dump_records.py out.npz [frames] [w h] [gop] [qp]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import _pkg  # noqa: E402

P = _pkg()
out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
gop = int(sys.argv[5]) if len(sys.argv) > 5 else 30
qp = int(sys.argv[6]) if len(sys.argv) > 6 else 26
ce = P.ClipEncoder(w, h, n, gop=gop, qp=qp, keep_records=1)
ce.generate_synth()
stream, sizes, st = ce.encode()
nmb = ((w + 15) // 16) * ((h + 15) // 16)
rec = np.empty((n, nmb, 2), np.int32)
ce.L.H264E_clip_read_records.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
for f in range(n):
    if ce.L.H264E_clip_read_records(ce.c, f, rec[f].ctypes.data):
        raise SystemExit("read_records(%d) failed" % f)
np.savez_compressed(out, rec=rec, w=w, h=h, gop=gop, qp=qp, relaunches=st.reencoded_gops)
print("%d frames, %d relaunches, %d bytes -> %s (%d bytes)" % (n, st.reencoded_gops, len(stream), out, os.path.getsize(out)))
