#!/usr/bin/env python3
"""Regenerate tests/golden/vbv.json: md5 / frame sizes of streams the REFERENCE encoder produces under frame-level rate control when
H264E_set_vbv_state (h264-lab.h:6898-6913) is called in the middle of the stream (oracle/vbv_harness.c: a translation unit that
includes the reference header and calls its public API; `make -C oracle vbv`).  Build container only; the output is data."""
import hashlib
import json
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "vbv_harness")

CASES = {
    # name: (w, h, frames, gop, kbps, [(frame, vbv_size_bytes, vbv_fullness_bytes or -1)])
    "cif_size_then_fullness": (352, 288, 40, 30, 500, [(10, 25000, -1), (20, 12500, 3000)]),
    "cif_level_changes_at_the_key_frame": (352, 288, 36, 12, 500, [(5, 500000, -1), (14, 500000, 20000), (30, 12500, 0)]),     # a larger VBV raises the SPS level at the next key frame
    "cif_overflow_transparent_frames": (352, 288, 24, 30, 300, [(6, 12500, 40000), (15, 6000, 30000)]),     # fullness far above the size: the reference codes transparent frames (h264-lab.h:6497-6510)
    "qcif_no_vbv": (176, 144, 20, 30, 200, [(4, 0, -1), (12, 12500, 1000)]),                                 # vbv_size 0 switches the VBV terms of the controller off
}


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "vbv"], stdout=subprocess.DEVNULL)
    out = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for name, (w, h, n, gop, kbps, ev) in CASES.items():
            o = os.path.join(tmp, "o.264")
            r = subprocess.run([HARNESS, str(w), str(h), str(n), str(gop), str(kbps), o] + ["%d:%d:%d" % e for e in ev], capture_output=True, text=True, check=True)
            data = open(o, "rb").read()
            sizes = [int(l.split("bytes=")[1]) for l in r.stdout.splitlines() if l.startswith("frame=")]
            assert len(sizes) == n and sum(sizes) == len(data)
            out[name] = dict(w=w, h=h, frames=n, gop=gop, kbps=kbps, events=[list(e) for e in ev], bytes=len(data), md5=hashlib.md5(data).hexdigest(), frame_bytes=sizes)
            print(name, len(data), out[name]["md5"], sizes[:8])
    json.dump(out, open(os.path.join(HERE, "vbv.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
