"""CPU: the product CLI (encode_app.c) linked against the test-only emulation library: the file pipeline of --clip 1 -- reader
thread, two staging buffers, input ring in "HBM", bounded output buffer, device-side PSNR sums -- on tiny clips with budgets
small enough that the ring wraps and the output buffer fills many times."""
import os
import subprocess

import pytest

import clips
import oracle_lib

HERE = os.path.dirname(os.path.abspath(__file__))
APP = os.path.join(HERE, "emu", "build", "encode_app_emu")


@pytest.fixture(scope="module", autouse=True)
def _emu():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "emu")], stdout=subprocess.DEVNULL)


@pytest.mark.parametrize("flags,kw", [("--qp 26 --gop 7", dict(gop=7, qp=26)), ("--qp 30 --gop 7 --threads 3", dict(gop=7, qp=30, slices=3)),
                                      ("--kbps 200 --gop 30", dict(gop=30, kbps=200))])
def test_clip_pipeline_bounded_buffers(tmp_path, flags, kw):
    w, h, n = 176, 144, 40
    c = clips.make("synth", w, h, n)
    yuv = tmp_path / ("app_%dx%d.yuv" % (w, h))
    c.tofile(yuv)
    fsz = w * h * 3 // 2
    want, sizes = oracle_lib.encode_clip(c, w, h, **kw)
    env = dict(os.environ, H264E_APP_STAGE_KB=str(fsz * 3 // 1024 + 1), H264E_APP_RING_KB=str(fsz * 12 // 1024), H264E_APP_OUT_KB="8")
    out = tmp_path / "o.264"
    r = subprocess.run([APP, "--input", str(yuv), "--output", str(out), "--clip", "1", "--stats", "x"] + flags.split(), env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert out.read_bytes() == want
    assert [l for l in r.stdout.splitlines() if l.startswith("frame=")] == ["frame=%d, bytes=%d" % (i, b) for i, b in enumerate(sizes)]
    assert "input ring 12 frames, staging 2 x 3 frames" in r.stderr


@pytest.mark.skipif(not os.path.exists(oracle_lib.REF_APP), reason="compiled reference (oracle/_ref) not present")
def test_psnr_line_equals_reference(tmp_path):
    """--psnr: same stdout line as the reference (minih264e_test.c:376-405), in clip mode from device-side sums of squared
    differences, in frame-at-a-time mode from the reconstruction written back into the input planes (h264-lab.h:6719-6723)"""
    w, h, n = 176, 144, 12
    c = clips.make("synth", w, h, n)
    yuv = tmp_path / ("app_%dx%d.yuv" % (w, h))
    c.tofile(yuv)
    flags = ["--qp", "26", "--gop", "5", "--psnr", "x"]
    ref = subprocess.run([oracle_lib.REF_APP, "--input", str(yuv), "--output", str(tmp_path / "r.264")] + flags, capture_output=True, text=True)
    line = [l for l in ref.stdout.splitlines() if "YPSNR" in l]
    assert len(line) == 1
    for extra in (["--clip", "0"], ["--clip", "1"], []):           # the frame-at-a-time loop, the clip encoder, the default (= clip)
        r = subprocess.run([APP, "--input", str(yuv), "--output", str(tmp_path / "o.264")] + flags + extra, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert [l for l in r.stdout.splitlines() if "YPSNR" in l] == line
        assert (tmp_path / "o.264").read_bytes() == (tmp_path / "r.264").read_bytes()


@pytest.mark.parametrize("name,w,h,n,flags,kw,gpus,redo", [
    ("synth", 176, 144, 40, "--qp 26 --gop 7", dict(gop=7, qp=26), 3, False),
    ("pan", 352, 288, 12, "--qp 26 --gop 3", dict(gop=3, qp=26), 2, True),
    ("pan", 352, 288, 12, "--qp 26 --gop 3 --threads 4", dict(gop=3, qp=26, slices=4), 4, False)])
def test_gpus_option_shards_one_stream(tmp_path, name, w, h, n, flags, kw, gpus, redo):
    """--gpus N: contiguous GOP blocks of the file on N clip encoders (all on the one emulated device here), started from a
    speculated mv_clusters state, settled in stream order with the exact state (8 bytes per boundary) -- the concatenation is the
    oracle's single stream; the fast pan defeats the speculation (frames are encoded again), row-band slices need none"""
    c = clips.make(name, w, h, n)
    yuv = tmp_path / ("app_%dx%d.yuv" % (w, h))
    c.tofile(yuv)
    fsz = w * h * 3 // 2
    want, sizes = oracle_lib.encode_clip(c, w, h, **kw)
    env = dict(os.environ, H264E_APP_STAGE_KB=str(fsz * 2 // 1024 + 1), H264E_APP_RING_KB=str(fsz * 8 // 1024), H264E_APP_OUT_KB="8")
    out = tmp_path / "o.264"
    r = subprocess.run([APP, "--input", str(yuv), "--output", str(out), "--clip", "1", "--stats", "x", "--gpus", str(gpus)] + flags.split(), env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert out.read_bytes() == want
    assert [l for l in r.stdout.splitlines() if l.startswith("frame=")] == ["frame=%d, bytes=%d" % (i, b) for i, b in enumerate(sizes)]
    last = r.stderr.strip().splitlines()[-1]
    assert ("GOP-sharded over %d clip encoders" % gpus) in last
    assert (" 0 frames encoded again" not in last) == redo


def test_refused_options_empty_input_and_upload_failure(tmp_path):
    """the CLI's error paths, none of which may hang or write a different stream: options of the reference that this encoder
    refuses (minih264e_test.c:135 --gen, :163 --denoise) -> exit 1 and no output file; an empty input and --gop 0 -> an empty
    stream and exit 0 like the reference's read loop (minih264e_test.c:654); a failed asynchronous upload in the clip pipeline
    (injected) -> error text and a non-zero exit in bounded time; a non-seekable input -> the frame-at-a-time loop"""
    w, h, n = 176, 144, 6
    c = clips.make("synth", w, h, n)
    yuv = tmp_path / ("app_%dx%d.yuv" % (w, h))
    c.tofile(yuv)
    out = tmp_path / "o.264"
    for opt in ("--denoise", "--gen"):
        r = subprocess.run([APP, "--input", str(yuv), "--output", str(out), opt, "x", "--qp", "26"], capture_output=True, text=True, timeout=60)
        assert r.returncode == 1 and "not supported" in r.stdout and not out.exists()
    empty = tmp_path / ("empty_%dx%d.yuv" % (w, h))
    empty.write_bytes(b"")
    for flags in (["--gop", "0"], ["--gop", "5"], ["--gop", "0", "--clip", "0"]):
        r = subprocess.run([APP, "--input", str(empty), "--output", str(out)] + flags, capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and out.read_bytes() == b"", r.stdout + r.stderr
    want, _ = oracle_lib.encode_clip(c, w, h, gop=0, qp=26)
    r = subprocess.run([APP, "--input", str(yuv), "--output", str(out), "--gop", "0", "--qp", "26"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and out.read_bytes() == want
    # the second asynchronous upload fails: the run ends with the device layer's message instead of waiting for frames forever
    fsz = w * h * 3 // 2
    env = dict(os.environ, H264E_TEST_KNOBS="1", H264E_TEST_UPLOAD_FAIL_AT="1", H264E_APP_STAGE_KB=str(fsz * 2 // 1024 + 1), H264E_APP_RING_KB=str(fsz * 8 // 1024))
    r = subprocess.run([APP, "--input", str(yuv), "--output", str(out), "--qp", "26"], env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "upload failed" in r.stdout and "injected failure" in r.stdout, r.stdout + r.stderr
    # without the explicit switch the knob is ignored
    env.pop("H264E_TEST_KNOBS")
    r = subprocess.run([APP, "--input", str(yuv), "--output", str(out), "--qp", "26", "--gop", "30"], env=env, capture_output=True, text=True, timeout=120)
    want30, _ = oracle_lib.encode_clip(c, w, h, gop=30, qp=26)
    assert r.returncode == 0 and out.read_bytes() == want30
    # a pipe has no length: the reference's read-until-EOF loop takes over
    with open(yuv, "rb") as f:
        r = subprocess.run("cat | %s --input /dev/stdin_%dx%d --output %s --qp 26 --gop 30" % (APP, w, h, out), shell=True, stdin=f, capture_output=True, text=True, timeout=120)
    assert "cant open input file" in r.stdout      # the size lives in the file NAME (guess_format): /dev/stdin_176x144 does not exist
    link = tmp_path / ("pipe_%dx%d.yuv" % (w, h))
    os.mkfifo(link)
    p = subprocess.Popen([APP, "--input", str(link), "--output", str(out), "--qp", "26", "--gop", "30"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    with open(link, "wb") as f:
        f.write(c.tobytes())
    so, se = p.communicate(timeout=120)
    assert p.returncode == 0 and out.read_bytes() == want30, so.decode() + se.decode()
