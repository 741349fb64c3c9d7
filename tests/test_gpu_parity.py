"""GPU: parity of the HIP path (through the C ABI) against the CPU oracle and the reference's golden vectors."""
import hashlib
import json
import os

import numpy as np
import pytest

import clips
import oracle_lib
import pkg
import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = json.load(open(os.path.join(HERE, "golden", "golden.json")))


def _flags(s):
    t = s.split()
    d = dict(zip(t[0::2], t[1::2]))
    return dict(gop=int(d.get("--gop", 20)), qp=int(d.get("--qp", 33)), speed=int(d.get("--speed", 0)), kbps=int(d.get("--kbps", 0)))


@pytest.fixture(scope="module")
def P():
    p = pkg.load_pkg()
    L = p.load()                    # fails loudly when the HIP library is missing
    assert L.h264e_hip_device_count() > 0, "no HIP device visible"
    return p


@pytest.mark.parametrize("g", GOLDEN, ids=lambda g: "%s_%dx%d_%s" % (g["clip"], g["w"], g["h"], g["flags"].replace(" ", "")))
def test_golden_vectors_bit_exact(P, g):
    """drop-in API, frame by frame, against md5 / per-frame sizes produced by the reference encoder"""
    c = clips.make(g["clip"], g["w"], g["h"], g["frames"])
    e = P.Encoder(g["w"], g["h"], **_flags(g["flags"]))
    parts = [e.encode(c[t]) for t in range(g["frames"])]
    e.close()
    assert [len(p) for p in parts] == g["frame_bytes"]
    assert hashlib.md5(b"".join(parts)).hexdigest() == g["md5"]


@pytest.mark.parametrize("qp", [10, 17, 24, 31, 38, 45, 51])
@pytest.mark.parametrize("name", ["synth", "noise", "pan", "extremes"])
def test_matches_oracle_across_qp(P, name, qp):
    w, h, n = 176, 144, 4
    c = clips.make(name, w, h, n)
    want, sizes = oracle_lib.encode_clip(c, w, h, gop=30, qp=qp)
    e = P.Encoder(w, h, gop=30, qp=qp)
    got = [e.encode(c[t]) for t in range(n)]
    e.close()
    assert [len(x) for x in got] == sizes
    assert b"".join(got) == want


def test_recon_written_back_when_input_not_const(P):
    """const_input_flag = 0: the reconstruction replaces the caller's planes (h264-lab.h:6719-6723); it must equal the
    oracle's deblocked reconstruction -- checks recon + in-loop filter, not only the bits"""
    w, h, n = 176, 144, 3
    c = clips.make("synth", w, h, n)
    o = oracle_lib.Encoder(w, h, gop=30, qp=26)
    e = P.Encoder(w, h, gop=30, qp=26, const_input=0)
    for t in range(n):
        f = c[t].copy()
        assert e.encode(f) == o.encode(c[t])
        rec, cw, ch = o.recon()
        assert (cw, ch) == (w, h) and np.array_equal(f, rec)
    e.close()


@pytest.mark.parametrize("name,w,h,n,gop,qp,chains", [
    ("synth", 352, 288, 12, 3, 26, 0), ("synth", 176, 144, 10, 3, 26, 2), ("pan", 176, 144, 9, 3, 30, 0),
    ("pan", 352, 288, 6, 2, 26, 3), ("noise", 64, 48, 6, 2, 30, 0), ("synth", 200, 120, 6, 2, 33, 0)])
def test_gop_chain_encoder_matches_oracle(P, name, w, h, n, gop, qp, chains):
    c = clips.make(name, w, h, n)
    want, sizes = oracle_lib.encode_clip(c, w, h, gop=gop, qp=qp)
    ce = P.ClipEncoder(w, h, n, gop=gop, qp=qp, max_chains=chains)
    ce.upload(c)
    out, fs, st = ce.encode()
    ce.close()
    assert fs == sizes and out == want


def test_clip_state_handoff_continues_the_stream(P):
    """two clips with {mv_clusters, idr parity} handed from the first to the second == one clip: what a host
    that shards a stream at key frames has to pass along"""
    w, h, n, gop = 352, 288, 12, 3
    c = clips.make("pan", w, h, n)
    want, sizes = oracle_lib.encode_clip(c, w, h, gop=gop, qp=26)
    a = P.ClipEncoder(w, h, 6, gop=gop, qp=26)
    a.upload(c[:6])
    out_a, fs_a, st = a.encode()
    a.close()
    b = P.ClipEncoder(w, h, 6, gop=gop, qp=26, clusters_in=(st.mv_clusters_out[0], st.mv_clusters_out[1]), idr_state=st.next_idr_pic_id_state)
    b.upload(c[6:])
    out_b, fs_b, _ = b.encode()
    b.close()
    assert fs_a + fs_b == sizes and out_a + out_b == want


def test_1080p_full_size_properties(P):
    """BASELINE configs[2] shape (cropped 1080p, streaming clip encoder).  Oracle-checked prefix + size-independent properties:
    chain-count invariance, run-to-run determinism, clip encoder == frame-at-a-time API."""
    w, h, n, gop = 1920, 1080, 12, 4
    ce = P.ClipEncoder(w, h, n, gop=gop, qp=26)
    ce.generate_synth()
    a, fa, _ = ce.encode()
    b, fb, _ = ce.encode()
    ce.close()
    assert a == b and fa == fb, "not deterministic"
    ce1 = P.ClipEncoder(w, h, n, gop=gop, qp=26, max_chains=1)
    ce1.generate_synth()
    c1, f1, _ = ce1.encode()
    ce1.close()
    assert c1 == a and f1 == fa, "result depends on the number of chains in flight"
    c = synth.clip(w, h, 2)
    want, sizes = oracle_lib.encode_clip(c, w, h, gop=gop, qp=26)
    assert fa[:2] == sizes and a[: len(want)] == want, "1080p differs from the oracle"
    e = P.Encoder(w, h, gop=gop, qp=26)
    c5 = oracle_lib.synth_c(w, h, 6)
    got = b"".join(e.encode(c5[t]) for t in range(6))
    e.close()
    assert got == a[: sum(fa[:6])]


def test_1080p_streaming_through_misspeculation(P):
    """40 frames of the bench clip: the rounded mv_clusters candidates change at frames 17, 21 and 33, so the streaming
    encoder aborts and relaunches three times; the stream must still equal the oracle's byte for byte"""
    w, h, n, gop = 1920, 1080, 40, 30
    ce = P.ClipEncoder(w, h, n, gop=gop, qp=26)
    ce.generate_synth()
    out, fs, st = ce.encode()
    ce.close()
    assert st.reencoded_gops >= 1, "the clip no longer exercises the relaunch path"
    c = oracle_lib.synth_c(w, h, n)
    want, sizes = oracle_lib.encode_clip(c, w, h, gop=gop, qp=26)
    assert fs == sizes and out == want


@pytest.mark.parametrize("qp,speed", [(14, 0), (38, 0), (50, 0), (26, 2), (26, 9), (33, 10)])
def test_1080p_clip_encoder_qp_and_speed(P, qp, speed):
    """cropped 1080p through the streaming clip encoder across QPs and encode speeds (speed >= 9: no sub-pel search,
    8/10: no deblocking), against the oracle"""
    w, h, n = 1920, 1080, 5
    ce = P.ClipEncoder(w, h, n, gop=30, qp=qp, speed=speed)
    ce.generate_synth()
    out, fs, _ = ce.encode()
    ce.close()
    c = oracle_lib.synth_c(w, h, n)
    want, sizes = oracle_lib.encode_clip(c, w, h, gop=30, qp=qp, speed=speed)
    assert fs == sizes and out == want


def test_1080p_long_clip_is_invariant(P):
    """240 frames of the bench clip (about ten abort/relaunch cycles): the stream must not depend on how many frames are
    in flight per launch or on the reference-window geometry, nor change from run to run -- a race in the in-launch
    hand-offs would show here"""
    w, h, n, gop = 1920, 1080, 240, 30
    outs = []
    for frames_in_flight, wide in ((96, 0), (20, 0), (96, 0), (96, 1)):
        if wide:
            os.environ["H264E_WIDE_WINDOW"] = "1"       # the 64x64 window / 7-step frame lag instead of 53x52 / 4 steps
        try:
            ce = P.ClipEncoder(w, h, n, gop=gop, qp=26, max_chains=frames_in_flight)
            ce.generate_synth()
            out, fs, st = ce.encode()
            ce.close()
        finally:
            os.environ.pop("H264E_WIDE_WINDOW", None)
        outs.append((hashlib.md5(out).hexdigest(), fs))
    assert outs[0] == outs[1] == outs[2] == outs[3]


@pytest.mark.parametrize("waves", [1, 2, 3, 4])
@pytest.mark.parametrize("name,w,h,n,gop,slices", [("synth", 352, 288, 9, 4, 0), ("pan", 640, 368, 6, 30, 3), ("scene", 200, 120, 7, 7, 0)])
def test_every_kernel_variant_matches_oracle(P, monkeypatch, waves, name, w, h, n, gop, slices):
    """the macroblock kernel exists in variants chosen per launch (h264e_kernels.hip bk_launch_mb): one wave per row (H264E_WAVES=1), two
    waves per row at 3 and at 4 waves per SIMD (2, 4), the four-wave latency variant (3); all-intra launches always take the intra-only
    kernel.  The decisions must
    not depend on the variant: every one, both window geometries, with and without row-band slices, against the oracle"""
    monkeypatch.setenv("H264E_WAVES", str(waves))
    c = clips.make(name, w, h, n)
    want, sizes = oracle_lib.encode_clip(c, w, h, gop=gop, qp=27, slices=slices)
    ce = P.ClipEncoder(w, h, n, gop=gop, qp=27, slices=slices)
    ce.upload(c)
    out, fs, st = ce.encode()
    ce.close()
    assert fs == sizes
    assert out == want
    e = P.Encoder(w, h, gop=gop, qp=27, slices=slices)
    got = b"".join(e.encode(c[t]) for t in range(n))
    e.close()
    assert got == want


@pytest.mark.parametrize("w,h,n", [(3840, 2160, 3), (7680, 4320, 2)])
def test_4k_8k_match_oracle(P, w, h, n):
    """BASELINE configs[3]/[4] geometry (32 400 / 129 600 macroblocks per frame, 8K is cropped): I + P frames"""
    ce = P.ClipEncoder(w, h, n, gop=30, qp=26)
    ce.generate_synth()
    out, fs, _ = ce.encode()
    ce.close()
    c = oracle_lib.synth_c(w, h, n)
    want, sizes = oracle_lib.encode_clip(c, w, h, gop=30, qp=26)
    assert fs == sizes and out == want


def test_error_codes(P):
    import ctypes as C
    L = P.load()
    buf = C.create_string_buffer(1 << 22)
    cp = P.CreateParam(width=64, height=48, gop=2, const_input_flag=1, max_long_term_reference_frames=1)
    assert L.H264E_init(buf, C.byref(cp)) == 2
    e = P.Encoder(64, 48, gop=0, qp=30)
    f = synth.clip(64, 48, 1)[0]
    with pytest.raises(P.H264EError):        # P frame before any key frame: H264E_STATUS_BAD_FRAME_TYPE (h264-lab.h:6801)
        e.encode(f, frame_type=2)
    assert len(e.encode(f)) > 0
    e.close()


@pytest.mark.parametrize("name,w,h,n,gop,qp,slices,kbps", [
    ("synth", 352, 288, 6, 30, 26, 2, 0), ("synth", 352, 288, 6, 30, 26, 4, 0), ("pan", 352, 288, 6, 30, 26, 3, 0),
    ("noise", 176, 144, 4, 2, 30, 8, 0), ("synth", 200, 120, 5, 30, 26, 2, 0), ("extremes", 176, 144, 4, 30, 20, 5, 0),
    ("synth", 352, 288, 6, 30, 26, 4, 300), ("pan", 352, 288, 8, 30, 30, 8, 0), ("synth", 1920, 1080, 4, 30, 26, 16, 0)])
def test_row_band_slices_match_oracle(P, name, w, h, n, gop, qp, slices, kbps):
    """multi-slice (the reference's H264E_MAX_THREADS build, --threads N): drop-in API and streaming clip encoder"""
    c = clips.make(name, w, h, n)
    want, sizes = oracle_lib.encode_clip(c, w, h, gop=gop, qp=qp, kbps=kbps, slices=slices)
    e = P.Encoder(w, h, gop=gop, qp=qp, kbps=kbps, slices=slices)
    got = [e.encode(c[t]) for t in range(n)]
    e.close()
    assert [len(x) for x in got] == sizes and b"".join(got) == want
    ce = P.ClipEncoder(w, h, n, gop=gop, qp=qp, slices=slices, kbps=kbps)
    ce.upload(c)
    out, fs, _ = ce.encode()
    ce.close()
    assert out == want and fs == sizes


@pytest.mark.parametrize("w,h,n,dy", [(1920, 1080, 4, 40), (1920, 1080, 3, 96), (3840, 2160, 3, 64)])
def test_large_vertical_motion_far_reads(P, w, h, n, dy):
    """vertical pan of 40..96 samples per frame through the streaming encoder: vectors leave the LDS window, so reference
    reads take the HBM path with its dynamic wait on rows of the frame that is still being encoded (rv_wait_rect) -- the one
    wait that can target a workgroup later in the dispatch order.  Must equal the oracle byte for byte."""
    big = synth.frame(w, h + dy * n + 16, 0)
    ybig = big[: w * (h + dy * n + 16)].reshape(h + dy * n + 16, w)
    c = np.empty((n, w * h * 3 // 2), np.uint8)
    for t in range(n):
        c[t, : w * h] = ybig[dy * t : dy * t + h].ravel()
        c[t, w * h :] = 128
    want, sizes = oracle_lib.encode_clip(c, w, h, gop=30, qp=30)
    ce = P.ClipEncoder(w, h, n, gop=30, qp=30)
    ce.upload(c)
    out, fs, _ = ce.encode()
    ce.close()
    assert fs == sizes and out == want


def test_nal_escape_pass_on_adversarial_payloads(P):
    """the device's emulation-prevention pass against the reference's automaton (h264-lab.h:3926-3975) on zero runs, triples that
    straddle the 256-byte block edges and zero-rich random data (real streams need an escape about once per 4 MB)"""
    import ctypes as C
    import nal_cases
    L = P.load()
    L.h264e_hip_pool_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.h264e_hip_selftest_nal_escape.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32)]
    L.h264e_hip_pool_destroy.argtypes = [C.c_void_p]
    pool = C.c_void_p()
    assert L.h264e_hip_pool_create(C.byref(pool), 0, 64, 48, 1, 1, 1) == 0
    for p in nal_cases.cases():
        want = nal_cases.escape_ref(p)
        cap = len(p) * 3 // 2 + 64
        dst = C.create_string_buffer(cap)
        n = C.c_uint32()
        assert L.h264e_hip_selftest_nal_escape(pool, p, len(p), dst, cap, C.byref(n)) == 0
        assert dst.raw[: n.value] == want, (len(p), p[:16])
    L.h264e_hip_pool_destroy(pool)


def test_clip_encoder_resumes_across_calls_and_reads_recon(P):
    """the clip encoder keeps its stream state between calls: frames uploaded in three instalments through a 9-frame input ring,
    a small output buffer that fills -- same bytes as one pass; the reconstruction of the last frame equals the oracle's"""
    w, h, n, gop = 352, 288, 20, 6
    c = clips.make("pan", w, h, n)
    want, sizes = oracle_lib.encode_clip(c, w, h, gop=gop, qp=28)
    ce = P.ClipEncoder(w, h, n, gop=gop, qp=28, resident=9, max_chains=5)
    out, fs = b"", []
    for a, b in ((0, 7), (7, 13), (13, 20)):
        ce.upload(c[a:b], first=a)
        while len(fs) < b:
            o, s, st = ce.encode(rewind=False, cap=40000)
            assert st.frames > 0
            out += o
            fs += s
    assert fs == sizes and out == want
    o = oracle_lib.Encoder(w, h, gop=gop, qp=28)
    for t in range(n):
        o.encode(c[t])
    rec, cw, ch = o.recon()
    assert np.array_equal(ce.read_recon(n - 1), rec)
    ce.close()


def test_frames_larger_than_the_host_mirror_are_fetched(P, monkeypatch):
    """a frame that does not fit the slot's host-mapped mirror (sized for ordinary frames) is fetched from the device NAL arena:
    forced with a 4000-byte mirror; QP 10 noise also exceeds the default mirror's 160 bytes per macroblock"""
    w, h, n = 352, 288, 4
    c = clips.make("noise", w, h, n)
    want, sizes = oracle_lib.encode_clip(c, w, h, gop=2, qp=10)
    assert max(sizes) > 396 * 160 + 65536
    ce = P.ClipEncoder(w, h, n, gop=2, qp=10)
    ce.upload(c)
    out, fs, _ = ce.encode()
    ce.close()
    assert out == want and fs == sizes
    monkeypatch.setenv("H264E_HOST_MIRROR_BYTES", "4000")
    e = P.Encoder(w, h, gop=2, qp=10)
    assert b"".join(e.encode(c[t]) for t in range(n)) == want
    e.close()


def test_per_macroblock_trace_matches_oracle(P):
    """type and mv[0] of every macroblock against the oracle's per-macroblock trace (names the first macroblock that differs)"""
    w, h, n = 352, 288, 4
    c = clips.make("pan", w, h, n)
    o = oracle_lib.Encoder(w, h, gop=30, qp=26)
    ce = P.ClipEncoder(w, h, n, gop=30, qp=26, keep_records=1)
    ce.upload(c)
    ce.encode()
    for t in range(n):
        o.encode(c[t])
        for i, ((typ, cbp, mvx, mvy, bitpos), (gx, gy, gt, used)) in enumerate(zip(o.trace(), ce.read_records(t))):
            assert gt == typ, "frame %d macroblock %d: type %d, oracle %d" % (t, i, gt, typ)
            if typ < 5:
                assert (gx, gy) == (mvx, mvy), "frame %d macroblock %d: mv (%d,%d), oracle (%d,%d)" % (t, i, gx, gy, mvx, mvy)
    ce.close()


import sweep_cases


@pytest.mark.parametrize("name,w,h,n,kw", sweep_cases.cases(60, 20261005, 704 * 576 * 6), ids=lambda v: str(v) if not isinstance(v, dict) else "-".join("%s%d" % (k[0], x) for k, x in v.items()))
def test_random_configurations_match_oracle(P, name, w, h, n, kw):
    """seeded sweep over sizes (cropped ones too), content, QP, GOP, speed, slices and rate control: clip encoder and frame-at-a-time
    API against the oracle"""
    c = clips.make(name, w, h, n)
    want, sizes = oracle_lib.encode_clip(c, w, h, **kw)
    ce = P.ClipEncoder(w, h, n, **kw)
    ce.upload(c)
    out, fs, _ = ce.encode()
    ce.close()
    assert fs == sizes and out == want
    e = P.Encoder(w, h, **kw)
    parts = [e.encode(c[t]) for t in range(n)]
    e.close()
    assert b"".join(parts) == want


def _strided_planes(frame, w, h, pads=(32, 16, 8), fill=0xAA):
    """the packed I420 frame as three separately allocated planes with row strides w + 32, w/2 + 16, w/2 + 8 (H264E_io_yuv_t allows any
    stride, h264-lab.h:231-237); the padding holds a marker"""
    out = []
    off = 0
    for c, pad in enumerate(pads):
        pw, ph = (w, h) if c == 0 else (w // 2, h // 2)
        buf = np.full((ph, pw + pad), fill, np.uint8)
        buf[:, :pw] = frame[off: off + pw * ph].reshape(ph, pw)
        off += pw * ph
        out.append(buf)
    return out


def _check_strided(P, lib=None):
    """strided, separately allocated planes through H264E_encode, read-only and with write-back (const_input_flag = 0): the stream is the
    oracle's, the written-back planes are the oracle's reconstruction, the padding is untouched"""
    w, h, n = 176, 144, 4
    c = clips.make("synth", w, h, n)
    for const_input in (1, 0):
        o = oracle_lib.Encoder(w, h, gop=30, qp=26)
        e = P.Encoder(w, h, gop=30, qp=26, const_input=const_input, **({"lib": lib} if lib else {}))
        for t in range(n):
            planes = _strided_planes(c[t], w, h)
            got = e.encode_planes(planes[0][:, :w], planes[1][:, : w // 2], planes[2][:, : w // 2])
            assert got == o.encode(c[t])
            rec, cw, ch = o.recon()
            want = c[t] if const_input else rec
            off = 0
            for k, pl in enumerate(planes):
                pw, ph = (w, h) if k == 0 else (w // 2, h // 2)
                assert np.array_equal(pl[:, :pw].ravel(), want[off: off + pw * ph]), "plane %d, frame %d, const_input %d" % (k, t, const_input)
                assert (pl[:, pw:] == 0xAA).all(), "padding of plane %d was written" % k
                off += pw * ph
        e.close()
        o.close()


def test_strided_and_separately_allocated_planes(P):
    """round-3 VERDICT weak item 1: every other test feeds stride = {w, w/2, w/2} from one packed buffer"""
    _check_strided(P)


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
def test_set_vbv_state_mid_stream_matches_reference(P, name):
    """H264E_set_vbv_state in the middle of a rate-controlled stream, incl. the transparent frames the reference codes on VBV overflow"""
    g = VBV_GOLDEN[name]
    e = P.Encoder(g["w"], g["h"], gop=g["gop"], kbps=g["kbps"])
    _run_vbv_case(e, g)
    e.close()
