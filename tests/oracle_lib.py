"""ctypes binding of the CPU oracle (oracle/build/liboracle_h264.so).  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "build", "liboracle_h264.so")
REF_APP = os.path.join(ROOT, "oracle", "_ref", "encode_app_ref")
REF_APP_THR = os.path.join(ROOT, "oracle", "_ref", "encode_app_ref_thr")       # -DH264E_MAX_THREADS=8 build: --threads N = N slices


class Param(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("width", "height", "gop", "qp", "speed", "vbv_size_bytes", "kbps", "slices")]


class Chain(C.Structure):
    _fields_ = [("mv_clusters", C.c_int32 * 2), ("next_idr_pic_id", C.c_int)]


class MbTrace(C.Structure):
    _fields_ = [("type", C.c_int8), ("cbp", C.c_uint8), ("mvx", C.c_int16), ("mvy", C.c_int16), ("bitpos", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB)
        _lib.h264o_open.restype = C.c_void_p
        _lib.h264o_open.argtypes = [C.POINTER(Param)]
        _lib.h264o_close.argtypes = [C.c_void_p]
        _lib.h264o_encode.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        _lib.h264o_encode_clip.restype = C.c_long
        _lib.h264o_encode_clip.argtypes = [C.POINTER(Param), C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]
        _lib.h264o_set_vbv_state.argtypes = [C.c_void_p, C.c_int, C.c_int]
        _lib.h264o_set_vbv_state.restype = None
        _lib.h264o_get_chain.argtypes = [C.c_void_p, C.POINTER(Chain)]
        _lib.h264o_set_chain.argtypes = [C.c_void_p, C.POINTER(Chain)]
        _lib.h264o_get_recon.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _lib.h264o_get_trace.restype = C.POINTER(MbTrace)
        _lib.h264o_get_trace.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        _lib.synth_v1_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint32]
    return _lib


def make_param(w, h, gop=30, qp=26, speed=0, kbps=0, slices=0):
    return Param(w, h, gop, qp, speed, 100000 // 8, kbps, slices)


def encode_clip(clip, w, h, gop=30, qp=26, speed=0, kbps=0, slices=0):
    """clip: uint8 array [nframes, w*h*3/2].  Returns (bitstream bytes, per-frame sizes)."""
    clip = np.ascontiguousarray(clip, dtype=np.uint8)
    n = clip.shape[0]
    cap = clip.size * 2 + (1 << 16)
    out = np.empty(cap, np.uint8)
    sizes = (C.c_int * n)()
    par = make_param(w, h, gop, qp, speed, kbps, slices)
    r = lib().h264o_encode_clip(C.byref(par), clip.ctypes.data, n, out.ctypes.data, cap, sizes)
    assert r >= 0
    return out[:r].tobytes(), list(sizes)


class Encoder:
    """Frame-at-a-time oracle encoder with access to recon, trace and the GOP hand-off state."""

    def __init__(self, w, h, gop=30, qp=26, speed=0, kbps=0, slices=0):
        self.w, self.h = w, h
        self.par = make_param(w, h, gop, qp, speed, kbps, slices)
        self.e = lib().h264o_open(C.byref(self.par))
        assert self.e

    def encode(self, frame):
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        w, h = self.w, self.h
        base = frame.ctypes.data
        yuv = (C.c_void_p * 3)(base, base + w * h, base + w * h * 5 // 4)
        st = (C.c_int * 3)(w, w // 2, w // 2)
        p, n = C.c_void_p(), C.c_int()
        lib().h264o_encode(self.e, yuv, st, C.byref(p), C.byref(n))
        return C.string_at(p, n.value)

    def set_vbv_state(self, vbv_size_bytes, vbv_fullness_bytes):
        lib().h264o_set_vbv_state(self.e, vbv_size_bytes, vbv_fullness_bytes)

    def recon(self):
        cw, ch = C.c_int(), C.c_int()
        lib().h264o_get_recon(self.e, None, C.byref(cw), C.byref(ch))
        buf = np.empty(cw.value * ch.value * 3 // 2, np.uint8)
        lib().h264o_get_recon(self.e, buf.ctypes.data, C.byref(cw), C.byref(ch))
        return buf, cw.value, ch.value

    def trace(self):
        n = C.c_int()
        t = lib().h264o_get_trace(self.e, C.byref(n))
        return [(t[i].type, t[i].cbp, t[i].mvx, t[i].mvy, t[i].bitpos) for i in range(n.value)]

    def get_chain(self):
        c = Chain()
        lib().h264o_get_chain(self.e, C.byref(c))
        return (c.mv_clusters[0], c.mv_clusters[1], c.next_idr_pic_id)

    def set_chain(self, st):
        c = Chain((C.c_int32 * 2)(st[0], st[1]), st[2])
        lib().h264o_set_chain(self.e, C.byref(c))

    def close(self):
        if self.e:
            lib().h264o_close(self.e)
            self.e = None

    def __del__(self):
        self.close()


def synth_c(w, h, n, seed=1):
    out = np.empty((n, w * h * 3 // 2), np.uint8)
    for t in range(n):
        lib().synth_v1_frame(out[t].ctypes.data, w, h, t, seed)
    return out
