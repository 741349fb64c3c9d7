"""One-off larger parity sweep on the GPU (not collected by pytest): python tests/sweep_gpu_run.py [cases] [seed].  300 seeded random
configurations (tests/sweep_cases.py) through the clip encoder against the oracle; round 2: 0 mismatches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sweep_cases, clips, oracle_lib, pkg
P = pkg.load_pkg()
bad = 0
cases = sweep_cases.cases(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 987654321, 704 * 576 * 5)
for i, (name, w, h, n, kw) in enumerate(cases):
    c = clips.make(name, w, h, n)
    want, sizes = oracle_lib.encode_clip(c, w, h, **kw)
    ce = P.ClipEncoder(w, h, n, **kw); ce.upload(c); out, fs, _ = ce.encode(); ce.close()
    if out != want or fs != sizes:
        bad += 1; print("MISMATCH", name, w, h, n, kw, flush=True)
print("sweep: %d cases, %d mismatches" % (len(cases), bad))
