"""GPU: the failure paths of the persistent kernel -- the only safety nets of a design whose forward progress rests on bounded
spins (h264e_kernels.hip): a producer that never publishes (spin expiry -> poisoned row counters -> errflag -> the finalizer gives up
-> an error, not a hang and not a stream), and a row bit buffer that overflows.  Fault injection through test-only environment
knobs read at pool creation (H264E_TEST_*)."""
import time

import pytest

import clips
import pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    p = pkg.load_pkg()
    assert p.load().h264e_hip_device_count() > 0, "no HIP device visible"
    return p


def test_stuck_producer_is_reported_not_hung(P, monkeypatch):
    """row 3 of the first frame exits without publishing: row 4 spins until the (shortened) bound expires, poisons its counter and
    raises the launch's error flag; every row below and the finalizer stop on the poison; the API returns an error in bounded time"""
    monkeypatch.setenv("H264E_TEST_KNOBS", "1")
    monkeypatch.setenv("H264E_TEST_STALL_ROW", "3")
    monkeypatch.setenv("H264E_TEST_SPIN_LIMIT", "20000")
    w, h, n = 352, 288, 4
    c = clips.make("synth", w, h, n)
    t0 = time.time()
    ce = P.ClipEncoder(w, h, n, gop=30, qp=26)
    ce.upload(c)
    with pytest.raises(P.H264EError) as ei:
        ce.encode()
    ce.close()
    assert time.time() - t0 < 60
    assert "gave up waiting" in str(ei.value) or "did not complete" in str(ei.value)
    e = P.Encoder(w, h, gop=30, qp=26)
    with pytest.raises(P.H264EError):
        e.encode(c[0])
    e.close()
    # and the library is usable afterwards
    monkeypatch.delenv("H264E_TEST_STALL_ROW")
    monkeypatch.delenv("H264E_TEST_SPIN_LIMIT")
    e = P.Encoder(w, h, gop=30, qp=26)
    assert len(e.encode(c[0])) > 1000
    e.close()


def test_row_bit_buffer_overflow_is_reported(P, monkeypatch):
    """a 16-byte-per-macroblock row bit buffer cannot hold QP 10 noise: the kernel flags the overflow instead of writing past the
    buffer, and both APIs fail with the reference-style error instead of returning a truncated stream"""
    monkeypatch.setenv("H264E_TEST_KNOBS", "1")
    monkeypatch.setenv("H264E_TEST_ROW_BYTES_PER_MB", "16")
    w, h, n = 176, 144, 2
    c = clips.make("noise", w, h, n)
    e = P.Encoder(w, h, gop=30, qp=10)
    with pytest.raises(P.H264EError) as ei:
        e.encode(c[0])
    e.close()
    assert "overflow" in str(ei.value)
    ce = P.ClipEncoder(w, h, n, gop=30, qp=10)
    ce.upload(c)
    with pytest.raises(P.H264EError) as ei:
        ce.encode()
    ce.close()
    assert "overflow" in str(ei.value)


def test_concurrent_clip_encoders_on_one_device_take_turns(P):
    """four host threads, four clip encoders, one GPU: launches of the persistent kernel must not run side by side (its
    forward-progress argument needs the device's wave slots to itself -- unguarded, this very test ended in 'bounded spin expired');
    the pools take turns through the per-device launch lock and every stream is still the golden one"""
    import hashlib
    import json
    import os
    import threading
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_big.json")))["bench_1080p_600"]
    w, h, n = g["w"], g["h"], 150
    encs = []
    for b in range(4):
        e = P.ClipEncoder(w, h, n, gop=30, qp=26)
        e.generate_synth(0, n, t0=0, seed=1)
        encs.append(e)
    outs, errs = [None] * 4, []

    def work(b):
        try:
            outs[b] = encs[b].encode()
        except Exception as ex:  # noqa: BLE001 -- reported below
            errs.append(repr(ex))

    for rep in range(2):
        th = [threading.Thread(target=work, args=(b,)) for b in range(4)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for out, sizes, st in outs:
            assert sizes == g["frame_bytes"][:n] and len(out) == sum(sizes)
    for e in encs:
        e.close()
    assert len(set(hashlib.md5(o[0]).hexdigest() for o in outs)) == 1


def test_second_process_on_the_same_device_fails_fast(P, tmp_path):
    """one encoder PROCESS per device (h264e_kernels.hip process guard: an advisory lock on a file named after the device's PCI bus id):
    while this process holds an encoder, encode_app started on the same GPU refuses with a message that names the holder; it runs
    once the encoder here is closed, and H264E_SHARE_DEVICE=1 overrides the guard"""
    import os
    import subprocess
    app = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "h264-lab_amd", "lib", "encode_app")
    w, h, n = 176, 144, 3
    c = clips.make("synth", w, h, n)
    yuv = tmp_path / ("g_%dx%d.yuv" % (w, h))
    c.tofile(yuv)
    out = tmp_path / "o.264"
    cmd = [app, "--input", str(yuv), "--output", str(out), "--qp", "26", "--gop", "30"]
    e = P.Encoder(w, h, gop=30, qp=26)
    want = b"".join(e.encode(c[t]) for t in range(n))
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "in use by another encoder process (pid %d)" % os.getpid() in r.stdout, r.stdout + r.stderr
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120, env=dict(os.environ, H264E_SHARE_DEVICE="1"))
    assert r.returncode == 0 and out.read_bytes() == want, r.stdout + r.stderr
    e.close()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and out.read_bytes() == want, r.stdout + r.stderr
