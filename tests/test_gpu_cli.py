"""GPU: the drop-in CLI (h264-lab_amd/lib/encode_app) against the reference's golden vectors and its option quirks
(reference minih264e_test.c:113-687)."""
import hashlib
import json
import os
import subprocess

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
    r = _run(["--input", str(yuv), "--output", str(out)] + g["flags"].split() + ["--stats", "x"], str(tmp_path))
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
