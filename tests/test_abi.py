"""CPU: the product library loads and exports every symbol include/*.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

import pkg

ROOT = pkg.ROOT


def _declared():
    names = set()
    for fn in ("h264e_mi355x.h", "h264e_hip.h"):
        src = open(os.path.join(ROOT, "include", fn)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(H264E_[a-z]\w*|h264e_hip_\w+)\s*\(", src))
    return sorted(n for n in names if not n.endswith("_t"))


def test_library_exports_declared_symbols():
    P = pkg.load_pkg()
    path = P.lib_path()
    if not os.path.exists(path):
        P.build()
    lib = ctypes.CDLL(path)
    decl = _declared()
    assert "H264E_encode" in decl and "h264e_hip_submit" in decl and len(decl) > 25
    missing = [n for n in decl if not hasattr(lib, n)]
    assert not missing, missing


def test_struct_layouts_match_reference_abi():
    # create 56 B (num_layers at 48), run 48 B (nalu_callback at 32), io_yuv 40 B
    P = pkg.load_pkg()
    assert ctypes.sizeof(P.CreateParam) == 56 and P.CreateParam.num_layers.offset == 48
    assert ctypes.sizeof(P.RunParam) == 48 and P.RunParam.nalu_callback.offset == 32
    assert ctypes.sizeof(P.IoYuv) == 40
    # the extension structs: the ctypes mirror against what the library was compiled with
    L = P.load()
    assert L.H264E_struct_size(0) == ctypes.sizeof(P.ClipParam)
    assert L.H264E_struct_size(1) == ctypes.sizeof(P.ClipStats)


def test_sizeof_and_parameter_errors_without_gpu():
    # H264E_sizeof does not touch the device: sizes and status codes of h264-lab.h:6252-6306
    P = pkg.load_pkg()
    L = P.load()
    sp, ss = ctypes.c_int(), ctypes.c_int()

    def call(**kw):
        cp = P.CreateParam(gop=30, vbv_size_bytes=12500, const_input_flag=1, num_layers=1, **kw)
        return L.H264E_sizeof(ctypes.byref(cp), ctypes.byref(sp), ctypes.byref(ss))

    assert call(width=352, height=288) == 0 and (sp.value, ss.value) == (369840, 239743)
    assert call(width=1920, height=1080) == 0 and (sp.value, ss.value) == (6559920, 4857727)
    assert call(width=3840, height=2160) == 0 and (sp.value, ss.value) == (25463472, 19263823)
    assert call(width=0, height=288) == 2
    assert call(width=353, height=288) == 5
    assert L.H264E_sizeof(None, ctypes.byref(sp), ctypes.byref(ss)) == 1
    cp = P.CreateParam(width=200, height=120, gop=1, const_input_flag=0)
    assert L.H264E_sizeof(ctypes.byref(cp), ctypes.byref(sp), ctypes.byref(ss)) == 4
