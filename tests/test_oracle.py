"""CPU tests of the oracle: synth generator, golden vectors from the reference, reference side-by-side."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import clips
import oracle_lib
import synth

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = json.load(open(os.path.join(HERE, "golden", "golden.json")))


def _flags(s):
    t = s.split()
    d = dict(zip(t[0::2], t[1::2]))
    return dict(gop=int(d.get("--gop", 20)), qp=int(d.get("--qp", 33)), speed=int(d.get("--speed", 0)), kbps=int(d.get("--kbps", 0)))


def test_synth_md5_appendix_a():
    # SURVEY.md Appendix A check values
    assert hashlib.md5(synth.clip(352, 288, 8).tobytes()).hexdigest() == "8868567eb2aaabe57037a522116aaf02"


def test_synth_numpy_equals_c():
    for w, h, n in [(64, 48, 3), (200, 120, 2), (352, 288, 2)]:
        assert np.array_equal(synth.clip(w, h, n), oracle_lib.synth_c(w, h, n))


@pytest.mark.parametrize("g", GOLDEN, ids=lambda g: "%s_%dx%d_%s" % (g["clip"], g["w"], g["h"], g["flags"].replace(" ", "")))
def test_oracle_matches_reference_golden(g):
    """bit-exact against outputs of the reference encoder itself (tests/golden/make_golden.py)"""
    c = clips.make(g["clip"], g["w"], g["h"], g["frames"])
    assert hashlib.md5(c.tobytes()).hexdigest() == g["input_md5"]
    data, sizes = oracle_lib.encode_clip(c, g["w"], g["h"], **_flags(g["flags"]))
    assert sizes == g["frame_bytes"]
    assert len(data) == g["bytes"]
    assert hashlib.md5(data).hexdigest() == g["md5"]
    if "stream" in g:
        assert data == open(os.path.join(HERE, "golden", g["stream"]), "rb").read()


@pytest.mark.skipif(not os.path.exists(oracle_lib.REF_APP), reason="compiled reference (oracle/_ref) not present")
@pytest.mark.parametrize("qp", [10, 18, 26, 34, 42, 51])
@pytest.mark.parametrize("mode", ["--gop 30", "--gop 2 --speed 1", "--gop 30 --speed 9"])
def test_oracle_vs_compiled_reference(tmp_path, qp, mode):
    """side-by-side with the reference binary on a fresh clip (not in the golden table)"""
    w, h, n = 128, 96, 5
    c = synth.clip(w, h, n, seed=qp)
    yuv = tmp_path / ("c_%dx%d.yuv" % (w, h))
    c.tofile(yuv)
    out = tmp_path / "o.264"
    flags = ("--qp %d " % qp) + mode
    subprocess.run([oracle_lib.REF_APP, "--input", str(yuv), "--output", str(out)] + flags.split(), check=True, capture_output=True)
    data, _ = oracle_lib.encode_clip(c, w, h, **_flags(flags))
    assert data == out.read_bytes()


def test_gop_chain_handoff():
    """a GOP encoded by a fresh encoder is identical to the in-stream GOP once
    {mv_clusters, idr parity} are handed over -- the only state that crosses a key frame in CQP."""
    w, h, gop, n = 176, 144, 3, 9
    c = clips.pan(w, h, n)
    whole, sizes = oracle_lib.encode_clip(c, w, h, gop=gop, qp=30)
    e = oracle_lib.Encoder(w, h, gop=gop, qp=30)
    states = []
    for t in range(n):
        if t % gop == 0:
            states.append(e.get_chain())
        e.encode(c[t])
    off = 0
    for k in range(n // gop):
        f = oracle_lib.Encoder(w, h, gop=gop, qp=30)
        f.set_chain(states[k])
        part = b"".join(f.encode(c[t]) for t in range(k * gop, (k + 1) * gop))
        ref = whole[off : off + sum(sizes[k * gop : (k + 1) * gop])]
        off += len(ref)
        assert part == ref, "GOP %d differs" % k


GOLDEN_BIG = json.load(open(os.path.join(HERE, "golden", "golden_big.json")))


def _flags_big(s):
    t = s.split()
    d = dict(zip(t[0::2], t[1::2]))
    return dict(gop=int(d.get("--gop", 20)), qp=int(d.get("--qp", 33)), speed=int(d.get("--speed", 0)), kbps=int(d.get("--kbps", 0)),
                slices=int(d.get("--threads", 0)))


# full-length streams of the REFERENCE (tests/golden/make_golden_big.py).  The CPU suite checks the ones the oracle finishes in
# seconds; the long ones (1080p x 600, 4K, 8K) are checked against the HIP path in tests/test_gpu_golden_big.py.
@pytest.mark.parametrize("name", ["cif_30_thr2", "cif_30_thr4", "cif_60_kbps", "cif_300_gop30", "cif_300_intra"])
def test_oracle_matches_reference_full_length(name):
    g = GOLDEN_BIG[name]
    c = oracle_lib.synth_c(g["w"], g["h"], g["frames"])
    assert hashlib.md5(c.tobytes()).hexdigest() == g["input_md5"]
    data, sizes = oracle_lib.encode_clip(c, g["w"], g["h"], **_flags_big(g["flags"]))
    assert sizes == g["frame_bytes"]
    assert hashlib.md5(data).hexdigest() == g["md5"]


@pytest.mark.skipif(not os.path.exists(oracle_lib.REF_APP_THR), reason="compiled multi-slice reference (oracle/_ref) not present")
@pytest.mark.parametrize("slices,flags", [(2, "--qp 26 --gop 30"), (3, "--qp 40 --gop 2"), (5, "--qp 18 --gop 30 --speed 2"), (8, "--kbps 200 --gop 30")])
def test_oracle_row_bands_vs_compiled_reference(tmp_path, slices, flags):
    """side-by-side with the reference built with -DH264E_MAX_THREADS=8 (oracle/_ref/encode_app_ref_thr) on a fresh clip"""
    w, h, n = 176, 144, 5
    c = clips.make("pan", w, h, n) if slices == 3 else synth.clip(w, h, n, seed=slices)
    yuv = tmp_path / ("c_%dx%d.yuv" % (w, h))
    c.tofile(yuv)
    out = tmp_path / "o.264"
    subprocess.run([oracle_lib.REF_APP_THR, "--input", str(yuv), "--output", str(out)] + flags.split() + ["--threads", str(slices)], check=True, capture_output=True)
    kw = _flags_big(flags)
    kw["slices"] = slices
    data, _ = oracle_lib.encode_clip(c, w, h, **kw)
    assert data == out.read_bytes()


VBV_GOLDEN = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vbv.json")))


def _run_vbv_case(enc, g):
    """frame-at-a-time encode under --kbps with H264E_set_vbv_state called where the fixture says (tests/golden/vbv.json: streams of the
    reference itself through oracle/vbv_harness.c, which calls the reference's own function)"""
    c = clips.make("synth", g["w"], g["h"], g["frames"])
    parts = []
    for t in range(g["frames"]):
        for at, size, full in g["events"]:
            if at == t:
                enc.set_vbv_state(size, full)
        parts.append(enc.encode(c[t]))
    assert [len(p) for p in parts] == g["frame_bytes"]
    assert hashlib.md5(b"".join(parts)).hexdigest() == g["md5"]


@pytest.mark.parametrize("name", sorted(VBV_GOLDEN))
def test_oracle_set_vbv_state_matches_reference(name):
    """the oracle's restatement of H264E_set_vbv_state (h264-lab.h:6898-6913) and of the transparent frame on VBV overflow (:6497-6510)"""
    g = VBV_GOLDEN[name]
    o = oracle_lib.Encoder(g["w"], g["h"], gop=g["gop"], kbps=g["kbps"])
    _run_vbv_case(o, g)
    o.close()
