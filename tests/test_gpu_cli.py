"""GPU: the drop-in CLI (h264-lab_amd/lib/encode_app) against the reference's golden vectors and its option quirks
(reference minih264e_test.c:113-687)."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

import clips
import pkg

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = json.load(open(os.path.join(HERE, "golden", "golden.json")))
APP = os.path.join(os.path.dirname(HERE), "h264-lab_amd", "lib", "encode_app")


def _run(args, cwd):
    return subprocess.run([APP] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)


@pytest.fixture(scope="module")
def app():
    pkg.load_pkg().load()            # fails loudly when the HIP library is missing
    assert os.path.exists(APP), "encode_app not built"
    return APP


@pytest.mark.parametrize("g", [g for g in GOLDEN if (g["w"], g["h"]) == (352, 288)][:6], ids=lambda g: g["flags"].replace(" ", ""))
def test_cli_matches_golden(app, tmp_path, g):
    """frame size parsed from the file name, reference stdout lines, byte-identical .264"""
    c = clips.make(g["clip"], g["w"], g["h"], g["frames"])
    yuv = tmp_path / ("clip_%dx%d.yuv" % (g["w"], g["h"]))
    yuv.write_bytes(c.tobytes())
    out = tmp_path / "o.264"
    for mode in ([], ["--clip", "0"]):           # the default (the streaming clip encoder) and the reference's loop over H264E_encode
        r = _run(["--input", str(yuv), "--output", str(out)] + g["flags"].split() + mode + ["--stats", "x"], str(tmp_path))
        text = r.stdout.decode()
        assert r.returncode == 0, text
        assert "sizeof_persist = 369840 sizeof_scratch = 239743" in text           # H264E_sizeof of the reference for CIF
        assert ["frame=%d, bytes=%d" % (i, b) for i, b in enumerate(g["frame_bytes"])] == [l for l in text.splitlines() if l.startswith("frame=")]
        assert hashlib.md5(out.read_bytes()).hexdigest() == g["md5"]


def test_cli_clip_mode_and_quirks(app, tmp_path):
    g = GOLDEN[0]
    c = clips.make(g["clip"], g["w"], g["h"], g["frames"])
    yuv = tmp_path / ("clip_%dx%d.yuv" % (g["w"], g["h"]))
    yuv.write_bytes(c.tobytes())
    out = tmp_path / "o.264"
    # streaming clip encoder behind the same CLI
    r = _run(["--input", str(yuv), "--output", str(out), "--clip", "1"] + g["flags"].split(), str(tmp_path))
    assert r.returncode == 0, r.stdout.decode()
    assert hashlib.md5(out.read_bytes()).hexdigest() == g["md5"]
    # every --long option swallows the next argv, flags included: "--stats --qp" leaves "26" as a bare argument -> error exit
    r = _run(["--input", str(yuv), "--output", str(out), "--stats", "--qp", "26"], str(tmp_path))
    assert r.returncode == 1 and b"Unknown option" in r.stdout
    # unopenable input
    r = _run(["--input", str(tmp_path / "missing_352x288.yuv"), "--output", str(out)], str(tmp_path))
    assert r.returncode == 1 and b"cant open input file" in r.stdout
    # options of the reference (minih264e_test.c:135, :163) that this encoder does not implement: refused, never a different stream
    for opt in ("--denoise", "--gen"):
        out.unlink(missing_ok=True)
        r = _run(["--input", str(yuv), "--output", str(out), opt, "x"] + g["flags"].split(), str(tmp_path))
        assert r.returncode == 1 and b"not supported" in r.stdout and not out.exists(), r.stdout.decode()


GOLDEN_BIG = json.load(open(os.path.join(HERE, "golden", "golden_big.json")))
SYNTH = os.path.join(os.path.dirname(HERE), "oracle", "build", "synth_v1")


def _synth_file(tmp_path, w, h, n):
    if not os.path.exists(SYNTH):
        subprocess.check_call(["make", "-C", os.path.join(os.path.dirname(HERE), "oracle"), "all"], stdout=subprocess.DEVNULL)
    yuv = tmp_path / ("sv1_%dx%d.yuv" % (w, h))
    subprocess.check_call([SYNTH, str(w), str(h), str(n), str(yuv)])
    return yuv


@pytest.mark.parametrize("name,extra", [("cif_300_gop30", []), ("cif_30_thr4", []), ("cif_60_kbps", []), ("1080p_30_thr8", [])])
def test_cli_clip_pipeline_full_length(app, tmp_path, name, extra):
    """--clip 1: the file streams through staging buffers and the HBM input ring (small budgets here, so both wrap) -- with
    --threads and --kbps too -- and the output equals the reference's full-length stream"""
    g = GOLDEN_BIG[name]
    yuv = _synth_file(tmp_path, g["w"], g["h"], g["frames"])
    fsz = g["w"] * g["h"] * 3 // 2
    env = dict(os.environ, H264E_APP_STAGE_KB=str(fsz * 7 // 1024 + 1), H264E_APP_RING_KB=str(fsz * 28 // 1024), H264E_APP_OUT_KB="512")
    out = tmp_path / "o.264"
    r = subprocess.run([APP, "--input", str(yuv), "--output", str(out), "--clip", "1", "--stats", "x"] + g["flags"].split() + extra, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    text = r.stdout.decode()
    assert r.returncode == 0, text + r.stderr.decode()
    assert hashlib.md5(out.read_bytes()).hexdigest() == g["md5"]
    assert [l for l in text.splitlines() if l.startswith("frame=")] == ["frame=%d, bytes=%d" % (i, b) for i, b in enumerate(g["frame_bytes"])]


def test_cli_4k_file_bounded_host_memory(app, tmp_path):
    """a 4K file through --clip 1 with the default budgets: output equals the reference's stream, and the process stays under
    2.5 GB of resident host memory (two pinned staging buffers, one output buffer, the slot ring's result mirrors: budgets, not file sizes)"""
    g = GOLDEN_BIG["4k_30"]
    yuv = _synth_file(tmp_path, g["w"], g["h"], g["frames"])
    out = tmp_path / "o.264"
    # peak resident memory of THIS run only: a fresh helper process runs the app as its single child and reports the child's
    # ru_maxrss (RUSAGE_CHILDREN of the test process itself is a maximum over every child any earlier test has waited for)
    helper = ("import resource, subprocess, sys\n"
              "r = subprocess.run(sys.argv[1:], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)\n"
              "sys.stdout.write(r.stdout.decode())\n"
              "print('MAXRSS_KB=%d RC=%d' % (resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss, r.returncode))\n")
    r = subprocess.run([sys.executable, "-c", helper, APP, "--input", str(yuv), "--output", str(out), "--clip", "1"] + g["flags"].split(),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = r.stdout.decode()
    tail = [l for l in text.splitlines() if l.startswith("MAXRSS_KB=")]
    assert r.returncode == 0 and tail and tail[-1].endswith("RC=0"), text
    assert hashlib.md5(out.read_bytes()).hexdigest() == g["md5"]
    rss_kb = int(tail[-1].split()[0].split("=")[1])
    # 2 x 384 MB staging + 64 MB output + <= 896 MB of host-mapped result mirrors (the slot ring's budget) + the HIP runtime itself
    # (code objects, three streams, signal pools: ~0.3 GB): about 2.1 GB whatever the length of the file
    assert rss_kb < 2.5 * 1024 * 1024, (rss_kb, text)


def test_cli_psnr_clip_mode_equals_frame_mode(app, tmp_path):
    """--psnr in clip mode (device-side sums of squared differences) prints the line the frame-at-a-time mode prints (which the
    CPU suite pins against the reference binary)"""
    g = GOLDEN[0]
    c = clips.make(g["clip"], g["w"], g["h"], g["frames"])
    yuv = tmp_path / ("clip_%dx%d.yuv" % (g["w"], g["h"]))
    yuv.write_bytes(c.tobytes())
    lines = []
    for extra in (["--clip", "0"], ["--clip", "1"], []):           # the frame-at-a-time loop, the clip encoder, the default (= clip)
        r = _run(["--input", str(yuv), "--output", str(tmp_path / "o.264")] + g["flags"].split() + ["--psnr", "x"] + extra, str(tmp_path))
        assert r.returncode == 0, r.stdout.decode()
        lines.append([l for l in r.stdout.decode().splitlines() if "YPSNR" in l])
        assert hashlib.md5((tmp_path / "o.264").read_bytes()).hexdigest() == g["md5"]
    assert lines[0] == lines[1] and len(lines[0]) == 1
    assert "YPSNR=40.84 db  UPSNR=45.54 db  VPSNR=46.88 db" in lines[0][0]      # SURVEY.md Appendix B, the reference's own line


def test_cli_psnr_with_rate_control_clip_equals_frame_mode(app, tmp_path):
    """--kbps --psnr: in clip mode the frames behind the first one of a launch run on a speculated QP, some of them as leaves in
    spare slots whose picture is moved into place afterwards -- stream, per-frame sizes and the PSNR line (computed on the device
    from those pictures) equal the frame-at-a-time run"""
    w, h, n = 352, 288, 40
    c = clips.make("scene", w, h, n)
    yuv = tmp_path / ("rc_%dx%d.yuv" % (w, h))
    yuv.write_bytes(c.tobytes())
    runs = []
    for extra in (["--clip", "0"], ["--clip", "1"], ["--clip", "1", "--threads", "3"], ["--clip", "0", "--threads", "3"]):
        out = tmp_path / "o.264"
        r = _run(["--input", str(yuv), "--output", str(out), "--kbps", "300", "--gop", "30", "--psnr", "x", "--stats", "x"] + extra, str(tmp_path))
        assert r.returncode == 0, r.stdout.decode()
        text = r.stdout.decode()
        runs.append((hashlib.md5(out.read_bytes()).hexdigest(), [l for l in text.splitlines() if l.startswith("frame=")], [l for l in text.splitlines() if "YPSNR" in l]))
    assert runs[0] == runs[1] and len(runs[0][1]) == n and len(runs[0][2]) == 1
    assert runs[2] == runs[3] and runs[2][0] != runs[0][0]


def test_cli_gpus_option_shards_one_stream(app, tmp_path):
    """--gpus 3 on the one GPU of this box (three clip encoders on device 0): GOP blocks of the 300-frame CIF stream, settled with
    the exact mv_clusters state in stream order; the output equals the reference's single stream"""
    g = GOLDEN_BIG["cif_300_gop30"]
    yuv = _synth_file(tmp_path, g["w"], g["h"], g["frames"])
    out = tmp_path / "o.264"
    r = subprocess.run([APP, "--input", str(yuv), "--output", str(out), "--clip", "1", "--gpus", "3"] + g["flags"].split(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stdout.decode() + r.stderr.decode()
    assert hashlib.md5(out.read_bytes()).hexdigest() == g["md5"]
    assert b"GOP-sharded over 3 clip encoders" in r.stderr


REF_DROPIN = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "encode_app_ref_dropin")


@pytest.mark.parametrize("g", [g for g in GOLDEN if (g["w"], g["h"]) == (352, 288)][:4], ids=lambda g: g["flags"].replace(" ", ""))
def test_reference_cli_linked_against_the_dropin(app, tmp_path, g):
    """the reference's OWN, unmodified minih264e_test.c, compiled against include/h264e_mi355x.h and linked with libh264e_mi355x.so in the
    build container (`make -C oracle dropin`, tests/test_dropin_link.py): the application the boundary exists for runs on the MI355X and
    writes the reference encoder's bytes and stdout lines (--psnr: its PSNR line is computed from the written-back reconstruction)"""
    if not os.path.exists(REF_DROPIN):
        pytest.skip("oracle/_ref/encode_app_ref_dropin not built (needs the reference sources: build container)")
    c = clips.make(g["clip"], g["w"], g["h"], g["frames"])
    yuv = tmp_path / ("clip_%dx%d.yuv" % (g["w"], g["h"]))
    yuv.write_bytes(c.tobytes())
    out = tmp_path / "o.264"
    r = subprocess.run([REF_DROPIN, "--input", str(yuv), "--output", str(out)] + g["flags"].split() + ["--stats", "x"], capture_output=True, timeout=300, cwd=str(tmp_path))
    text = r.stdout.decode()
    assert r.returncode == 0, text + r.stderr.decode()
    assert "sizeof_persist = 369840 sizeof_scratch = 239743" in text
    assert ["frame=%d, bytes=%d" % (i, b) for i, b in enumerate(g["frame_bytes"])] == [l for l in text.splitlines() if l.startswith("frame=")]
    assert hashlib.md5(out.read_bytes()).hexdigest() == g["md5"]
