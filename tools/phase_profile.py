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

NAMES = {0: "setup", 1: "load top+input", 2: "inter: predictors", 3: "inter: skip test", 4: "inter: candidates", 5: "inter: diamond full-pel",
         6: "inter: sub-pel", 7: "inter: partition loop rest", 8: "intra 16x16", 9: "intra 4x4", 10: "chroma prediction", 11: "mb_write (xform/quant/CAVLC/recon)",
         12: "ctx save + deblock + stores", 13: "WAIT for row above (poll+acquire)", 14: "publish (drain stores)"}


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
    t = (C.c_ulonglong * 32)()
    L.H264E_clip_stamps(ce.c, t)
    nmb = sum(t[20:23])
    tot = sum(t[i] for i in range(15))
    print("%dx%d %d frames gop %d: %d MBs (skip %d, inter %d, intra %d); mb kernel %.1f ms over %d launches" %
          (w, h, frames, gop, nmb, t[20], t[21], t[22], st.mb_kernel_ms, st.kernel_launches))
    print("%-42s %12s %8s %10s" % ("phase", "cycles/MB", "share", "us/MB@2.1G"))
    for i in range(15):
        print("%-42s %12.0f %7.1f%% %10.2f" % (NAMES[i], t[i] / nmb, 100.0 * t[i] / tot, t[i] / nmb / 2100.0))
    print("%-42s %12.0f %8s %10.2f" % ("total", tot / nmb, "", tot / nmb / 2100.0))
    if t[18]:
        print("diamond calls/MB %.2f  scan iterations/call %.1f  SAD batches/call %.2f  cycles/batch %.0f  cycles/diag-probe-phase per call %.0f" %
              (t[18] / nmb, t[19] / t[18], t[17] / t[18], t[16] / max(t[17], 1), t[23] / t[18]))
        if t[15]:
            print("diamond: entry..end of the full-pel search %.0f cycles per call (incl. batches and the diagonal probe)" % (t[15] / t[18]))
    if t[30]:
        print("mb_write: luma transform + quantiser + reconstruction %.0f cycles/MB (of the mb_write line above; the CAVLC of the residual blocks measured 2.3 k)" % (t[30] / nmb))
    if t[29]:
        print("effective shader clock over the rows' lifetimes: %.0f MHz (cycle counter / 100 MHz wall clock)" % (100.0 * t[28] / t[29]))
    if t[24] or t[25]:
        print("cycles per macroblock by type: skip %.0f  inter %.0f  intra %.0f;  partition set-up before each diamond call %.0f" %
              (t[24] / max(t[20], 1), t[25] / max(t[21], 1), t[26] / max(t[22], 1), t[27] / max(t[18], 1)))


if __name__ == "__main__":
    main()
