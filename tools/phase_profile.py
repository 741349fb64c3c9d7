#!/usr/bin/env python3
"""Per-phase cycle breakdown of the macroblock kernel (diagnostic -DH264E_STAMPS build).  Run on the GPU box:
   make -C h264-lab_amd/csrc stamps && python tools/phase_profile.py [frames] [w h]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["H264E_LIB"] = os.path.join(ROOT, "h264-lab_amd", "lib", "libh264e_mi355x_stamps.so")
from __graft_entry__ import _pkg  # noqa: E402

NAMES = {0: "S load input + window + records above", 1: "S setup", 2: "S inter: predictors", 3: "S inter: skip test", 4: "S inter: candidates", 5: "S inter: partition search: entry/exit", 24: "S   search: set-up per partition (predictor, range, start SAD)", 25: "S   search: full-pel scan", 26: "S   search: sub-pel", 27: "S   search: bookkeeping",
         7: "S inter: rest", 13: "S WAIT row above / reference frame (poll+acquire)", 23: "S wait: hand-off buffer free (R wave 2 MBs behind)", 6: "S wait: decision of x-1 (R wave)",
         17: "R WAIT row above (rest of its record: poll+acquire)", 15: "R wait: search wave's skip test", 8: "R intra 16x16", 9: "R intra 4x4: entry/exit", 19: "R   i4: per-block control (availability, contexts, cut-off)", 31: "R   i4: mode choice (9 predictions + SADs)", 18: "R   i4: transform / quantiser / reconstruction", 16: "R wait: inter decision", 10: "R merge + contexts + chroma prediction",
         11: "R mb_write (xform/quant/CAVLC/recon)", 12: "R ctx save + deblock + stores", 14: "publish / signal"}
ORDER = [13, 23, 0, 6, 1, 2, 3, 4, 5, 24, 25, 26, 27, 7, 17, 15, 8, 9, 19, 31, 18, 16, 10, 11, 12, 14]
SEARCH_SIDE = {13, 23, 0, 6, 1, 2, 3, 4, 5, 24, 25, 26, 27, 7}


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
    gop = int(sys.argv[4]) if len(sys.argv) > 4 else frames
    P = _pkg()
    ce = P.ClipEncoder(w, h, frames, gop=gop, qp=26)
    ce.generate_synth()
    out, sizes, st = ce.encode(profile=True)
    L = ce.L
    # the pool is private to the clip encoder: reach it through the stamp reader exported by the library
    L.H264E_clip_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    t = (C.c_ulonglong * 48)()
    L.H264E_clip_stamps(ce.c, t)
    nmb = sum(t[20:23])
    tot = sum(t[i] for i in ORDER)
    print("%dx%d %d frames gop %d: %d MBs (skip %d, inter %d, intra %d); mb kernel %.1f ms over %d launches; H264E_WAVES=%s" %
          (w, h, frames, gop, nmb, t[20], t[21], t[22], st.mb_kernel_ms, st.kernel_launches, os.environ.get("H264E_WAVES", "auto")))
    print("S = search side (with two waves per row: the search wave), R = reconstruction side (the reconstruction wave); with H264E_WAVES=3 (four waves per row)\nthe 8x8 search helper adds to the S search lines and the deblock/store wave owns 'ctx save + deblock + stores'")
    print("%-58s %12s %8s" % ("phase", "cycles/MB", "share"))
    for i in ORDER:
        print("%-58s %12.0f %7.1f%%" % (NAMES[i], t[i] / nmb, 100.0 * t[i] / tot))
    s_tot = sum(t[i] for i in ORDER if i in SEARCH_SIDE)
    print("%-58s %12.0f" % ("total search side", s_tot / nmb))
    print("%-58s %12.0f" % ("total reconstruction side", (tot - s_tot) / nmb))
    if t[24]:
        print("inside the partition search, cycles/MB: set-up per partition (predictor, range, start SAD) %.0f | full-pel scan %.0f | sub-pel %.0f | bookkeeping %.0f" %
              (t[24] / nmb, t[25] / nmb, t[26] / nmb, t[27] / nmb))
    if t[30]:
        print("mb_write: luma transform + quantiser + reconstruction %.0f cycles/MB (of the mb_write line above; the CAVLC of the residual blocks measured 2.3 k)" % (t[30] / nmb))
    if t[37]:
        nf = t[37]
        print("finalizer workgroups (%d frames), microseconds per frame: wait for the verdict of the frame in front %.1f | follow the frame's rows (wait; walk + splice of each completed row) %.1f | end of the walk %.1f | last row's splice + result record %.1f | NAL escaping + export %.1f" %
              (nf, t[33] / nf / 100.0, t[32] / nf / 100.0, t[34] / nf / 100.0, t[35] / nf / 100.0, t[36] / nf / 100.0))
    if t[29]:
        print("effective shader clock over the rows' lifetimes: %.0f MHz (cycle counter / 100 MHz wall clock)" % (100.0 * t[28] / t[29]))



if __name__ == "__main__":
    main()
