"""The drop-in boundary, checked with the reference's own application: /root/reference/src/minih264e_test.c, UNMODIFIED and compiled where
it lies, links against include/h264e_mi355x.h + libh264e_mi355x.so and imports nothing from it but the reference's three entry points
(recipe: `make -C oracle dropin`).  Build container only (the reference sources do not travel); the GPU box runs the
resulting binary in tests/test_gpu_cli.py::test_reference_cli_linked_against_the_dropin."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/src/minih264e_test.c"
APP = os.path.join(ROOT, "oracle", "_ref", "encode_app_ref_dropin")
LIB = os.path.join(ROOT, "h264-lab_amd", "lib", "libh264e_mi355x.so")


@pytest.mark.skipif(not os.path.exists(REF_SRC), reason="reference sources not present (GPU box)")
def test_reference_cli_links_against_the_dropin_library():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "h264-lab_amd", "csrc"), "all"], stdout=subprocess.DEVNULL)
    if os.path.exists(APP):
        os.remove(APP)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "dropin"], stdout=subprocess.DEVNULL)
    assert os.path.exists(APP), "the reference CLI did not build against include/h264e_mi355x.h"
    und = subprocess.run(["nm", "-u", APP], capture_output=True, text=True, check=True).stdout.split("\n")
    syms = sorted(l.split()[-1] for l in und if l.strip() and "@" not in l and not l.split()[-1].startswith(("_ITM", "__gmon", "__cxa")))
    assert syms == ["H264E_encode", "H264E_init", "H264E_sizeof"], syms
    # ... and those three come from the product library, which exports them
    exp = subprocess.run(["nm", "-D", "--defined-only", LIB], capture_output=True, text=True, check=True).stdout
    for s in syms:
        assert " T %s\n" % s in exp
    need = subprocess.run(["readelf", "-d", APP], capture_output=True, text=True, check=True).stdout
    assert "libh264e_mi355x.so" in need


@pytest.mark.skipif(not os.path.exists(REF_SRC), reason="reference sources not present (GPU box)")
def test_reference_cli_runs_on_the_emulated_dropin(tmp_path):
    """the same unmodified application linked against the EMULATION build of the library (the product's host code, kernels emulated on
    the CPU): it runs here, without a GPU, and writes the reference's bytes and stdout lines -- the boundary carries the application, not
    only its link step"""
    import hashlib
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import clips
    emu_dir = os.path.join(ROOT, "tests", "emu", "build")
    if not os.path.exists(os.path.join(emu_dir, "libh264e_emu.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "emu")], stdout=subprocess.DEVNULL)
    app = str(tmp_path / "ref_cli_emu")
    subprocess.check_call(["gcc", "-O2", "-w", "-DMINIH264_H", "-DMINIH264_IMPLEMENTATION_GUARD", "-include", os.path.join(ROOT, "include", "h264e_mi355x.h"),
                           "-o", app, REF_SRC, "-L" + emu_dir, "-lh264e_emu", "-Wl,-rpath," + emu_dir, "-lm"])
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
    for g in [g for g in golden if (g["w"], g["h"]) == (352, 288)][:3]:
        c = clips.make(g["clip"], g["w"], g["h"], g["frames"])
        yuv = tmp_path / ("clip_%dx%d.yuv" % (g["w"], g["h"]))
        yuv.write_bytes(c.tobytes())
        out = tmp_path / "o.264"
        r = subprocess.run([app, "--input", str(yuv), "--output", str(out)] + g["flags"].split() + ["--stats", "x"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "sizeof_persist = 369840 sizeof_scratch = 239743" in r.stdout
        assert ["frame=%d, bytes=%d" % (i, b) for i, b in enumerate(g["frame_bytes"])] == [l for l in r.stdout.splitlines() if l.startswith("frame=")]
        assert hashlib.md5(out.read_bytes()).hexdigest() == g["md5"]
