#!/usr/bin/env python3
"""Stability soak of the round-4 build (run on the GPU box): many passes of the configurations that exercise the progressive finalizer,
the padded XCD-band order, launch groups and rate-control trees; every pass must give the same bytes and no bounded wait may expire."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from __graft_entry__ import _pkg
P = _pkg()
def soak(w, h, n, gop, slices, kbps, passes):
    ce = P.ClipEncoder(w, h, n, gop=gop, qp=26, slices=slices, kbps=kbps); ce.generate_synth()
    hs = set(); spins = 0; t = time.time()
    for i in range(passes):
        out, fs, st = ce.encode(); hs.add(hashlib.md5(out).hexdigest()); spins += st.spin_relaunches
    ce.close()
    print("%dx%d x %d gop %d slices %d kbps %d: %d passes in %.1f s, %s, spin relaunches %d" % (w, h, n, gop, slices, kbps, passes, time.time() - t, "identical" if len(hs) == 1 else "DIFFER", spins), flush=True)
    return len(hs) == 1 and spins == 0
ok = True
ok &= soak(1920, 1080, 600, 30, 0, 0, 100)
ok &= soak(1920, 1080, 600, 30, 8, 0, 40)
ok &= soak(1920, 1080, 600, 30, 2, 0, 20)
ok &= soak(3840, 2160, 160, 30, 0, 0, 20)
ok &= soak(3840, 2160, 160, 30, 8, 0, 20)
ok &= soak(7680, 4320, 39, 30, 0, 0, 10)
ok &= soak(352, 288, 1000, 30, 0, 0, 30)
ok &= soak(1280, 720, 600, 30, 0, 0, 30)
ok &= soak(1920, 1080, 60, 30, 0, 4000, 40)
ok &= soak(1920, 1080, 60, 30, 8, 4000, 40)
# launch groups: 4 streams, 15 rounds
encs = []
for b in range(4):
    e = P.ClipEncoder(1920, 1080, 300, gop=30, qp=26); e.generate_synth(0, 300, t0=0, seed=1); encs.append(e)
hs = set(); t = time.time()
for i in range(15):
    outs = P.ClipEncoder.encode_multi(encs)
    for o in outs: hs.add(hashlib.md5(o[0]).hexdigest())
for e in encs: e.close()
print("launch group of 4 x 300 frames: 15 rounds in %.1f s, %s" % (time.time() - t, "identical" if len(hs) == 1 else "DIFFER"), flush=True)
ok &= len(hs) == 1
print("STRESS OK" if ok else "STRESS FAILED")
sys.exit(0 if ok else 1)
