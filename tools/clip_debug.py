#!/usr/bin/env python3
"""Streaming clip encoder timing with the per-launch timeline (H264E_DEBUG):  clip_debug.py [frames] [w h] [gop] [qp] [slices] [kbps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["H264E_DEBUG"] = "1"
from __graft_entry__ import _pkg  # noqa: E402

P = _pkg()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
gop = int(sys.argv[4]) if len(sys.argv) > 4 else 30
qp = int(sys.argv[5]) if len(sys.argv) > 5 else 26
slices = int(sys.argv[6]) if len(sys.argv) > 6 else 0
kbps = int(sys.argv[7]) if len(sys.argv) > 7 else 0
if os.environ.get("H264E_QUIET"):
    os.environ.pop("H264E_DEBUG", None)
ce = P.ClipEncoder(w, h, n, gop=gop, qp=qp, slices=slices, kbps=kbps)
ce.generate_synth()
ce.encode()                                     # warm-up (clock ramp, first-touch)
t = time.time()
out, fs, st = ce.encode(profile=True)
dt = time.time() - t
nmb = ((w + 15) // 16) * ((h + 15) // 16)
print("slices %d kbps %d: " % (slices, kbps), end="")
print("time %.2f s rounds %d reenc %d launches %d mb_ms %.1f enc_ms %.1f read_ms %.1f asm_ms %.1f  | %dx%d %d frames: %.2f M MB/s, %.1f fps, %d bytes" %
      (dt, st.rounds, st.reencoded_gops, st.kernel_launches, st.mb_kernel_ms, st.encode_ms, st.readback_ms, st.assemble_ms, w, h, n, n * nmb / dt / 1e6, n / dt, len(out)))
