#!/usr/bin/env python3
"""Throughput of B independent clip encoders running CONCURRENTLY on one GPU (one host thread, one pool and one HIP stream each).
A single stream is latency bound (a pass is ~35 frame latencies); several streams fill the refills of one
with the steady state of the others.

    python tools/multi_clip_probe.py [clips] [frames] [w] [h] [gop] [qp] [slices] [max_chains]
"""
import hashlib
import importlib.util
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_pkg():
    spec = importlib.util.spec_from_file_location("h264_lab_amd", os.path.join(ROOT, "h264-lab_amd", "__init__.py"),
                                                  submodule_search_locations=[os.path.join(ROOT, "h264-lab_amd")])
    m = importlib.util.module_from_spec(spec)
    sys.modules["h264_lab_amd"] = m
    spec.loader.exec_module(m)
    return m


def run(P, B, frames, w, h, gop, qp, slices, max_chains):
    encs = []
    for b in range(B):
        e = P.ClipEncoder(w, h, frames, gop=gop, qp=qp, speed=0, slices=slices, max_chains=max_chains)
        e.generate_synth(0, frames, t0=(137 * b if os.environ.get('PROBE_STAGGER') else 0), seed=1)
        encs.append(e)
    outs = [None] * B

    def work(b):
        outs[b] = encs[b].encode()

    best = None
    for rep in range(3):
        t0 = time.time()
        if os.environ.get("PROBE_THREADS"):
            # the old way: independent H264E_clip_encode calls on B threads (their launches take turns: the per-device launch lock)
            th = [threading.Thread(target=work, args=(b,)) for b in range(B)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        else:
            # H264E_clip_encode_multi: the clips' launches merged into one grid per round (launch group)
            outs = P.ClipEncoder.encode_multi(encs)
        dt = time.time() - t0
        best = dt if best is None or dt < best else best
    nmb = ((w + 15) // 16) * ((h + 15) // 16)
    md5s = sorted(set(hashlib.md5(o[0]).hexdigest() for o in outs))
    for e in encs:
        e.close()
    return dict(clips=B, seconds=best, value=B * frames * nmb / best, fps=B * frames / best, md5=md5s,
                relaunches=[o[2].reencoded_gops for o in outs])


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    B, frames, w, h, gop, qp, slices, mc = (a + [4, 600, 1920, 1080, 30, 26, 0, 0][len(a):])[:8]
    P = load_pkg()
    r = run(P, B, frames, w, h, gop, qp, slices, mc)
    print("%d clips x %d frames %dx%d slices %d chains %d: %.3f s, %.2f M MB/s aggregate, %.1f fps aggregate, md5 %s, relaunches %s" %
          (B, frames, w, h, slices, mc, r["seconds"], r["value"] / 1e6, r["fps"], r["md5"], r["relaunches"]), flush=True)
