"""ctypes mirror of include/h264e_mi355x.h (same names, argument meaning and status codes as the reference API,
/root/reference/src/h264-lab.h:83-312)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_DEFAULT_LIB = os.path.join(HERE, "lib", "libh264e_mi355x.so")

STATUS_SUCCESS, STATUS_BAD_ARGUMENT, STATUS_BAD_PARAMETER, STATUS_BAD_FRAME_TYPE = 0, 1, 2, 3
STATUS_SIZE_NOT_MULTIPLE_16, STATUS_SIZE_NOT_MULTIPLE_2 = 4, 5
FRAME_TYPE_DEFAULT, FRAME_TYPE_KEY, FRAME_TYPE_P = 0, 6, 2


class H264EError(RuntimeError):
    pass


class CreateParam(C.Structure):  # h264-lab.h:83-172 (H264E_SVC_API=1, H264E_MAX_THREADS=0): 56 bytes
    _fields_ = [(n, C.c_int) for n in (
        "width", "height", "gop", "vbv_size_bytes", "vbv_overflow_empty_frame_flag", "vbv_underflow_stuffing_flag",
        "fine_rate_control_flag", "const_input_flag", "max_long_term_reference_frames", "enableNEON",
        "temporal_denoise_flag", "sps_id", "num_layers", "inter_layer_pred_flag")]


NALU_CB = C.CFUNCTYPE(None, C.POINTER(C.c_ubyte), C.c_int, C.c_void_p)


class RunParam(C.Structure):  # h264-lab.h:177-226: 48 bytes
    _fields_ = [("encode_speed", C.c_int), ("frame_type", C.c_int), ("long_term_idx_use", C.c_int),
                ("long_term_idx_update", C.c_int), ("desired_frame_bytes", C.c_int), ("qp_min", C.c_int),
                ("qp_max", C.c_int), ("desired_nalu_bytes", C.c_int), ("nalu_callback", NALU_CB),
                ("nalu_callback_token", C.c_void_p)]


class IoYuv(C.Structure):  # h264-lab.h:231-237: 40 bytes
    _fields_ = [("yuv", C.c_void_p * 3), ("stride", C.c_int * 3)]


class ClipParam(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("width", "height", "gop", "qp", "speed", "vbv_size_bytes", "device", "max_chains",
                                       "first_idr_pic_id_state")] + [("mv_clusters_in", C.c_int32 * 2), ("slices", C.c_int), ("kbps", C.c_int), ("resident_frames", C.c_int), ("keep_records", C.c_int)]


class ClipStats(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("upload_ms", "encode_ms", "readback_ms", "assemble_ms", "mb_kernel_ms", "splice_kernel_ms")] + \
               [(n, C.c_int) for n in ("kernel_launches", "chains", "rounds", "reencoded_gops")] + \
               [("mv_clusters_out", C.c_int32 * 2), ("next_idr_pic_id_state", C.c_int), ("first_frame", C.c_int), ("frames", C.c_int), ("spin_relaunches", C.c_int),
                ("relaunches_timed", C.c_int), ("processed_mbs", C.c_longlong), ("delivered_mbs", C.c_longlong), ("first_frame_ms_after_relaunch", C.c_double)]


def lib_path():
    return os.environ.get("H264E_LIB", _DEFAULT_LIB)


def build():
    """Compile the HIP library for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-C", os.path.join(HERE, "csrc"), "all"], stdout=subprocess.DEVNULL)


_libs = {}


def load(path=None):
    path = path or lib_path()
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise H264EError("HIP library %s not built: run `make -C h264-lab_amd/csrc` (there is no CPU fallback)" % path)
    L = C.CDLL(path)
    L.H264E_sizeof.argtypes = [C.POINTER(CreateParam), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.H264E_init.argtypes = [C.c_void_p, C.POINTER(CreateParam)]
    L.H264E_encode.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(RunParam), C.POINTER(IoYuv), C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
    L.H264E_set_vbv_state.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.H264E_set_vbv_state.restype = None
    L.H264E_set_slices.argtypes = [C.c_void_p, C.c_int]
    L.H264E_close.argtypes = [C.c_void_p]
    L.H264E_close.restype = None
    L.H264E_set_device.argtypes = [C.c_int]
    L.H264E_set_device.restype = None
    L.H264E_last_error.restype = C.c_char_p
    L.H264E_clip_open.argtypes = [C.POINTER(C.c_void_p), C.POINTER(ClipParam), C.c_int]
    L.H264E_clip_upload.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.H264E_clip_generate_synth.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint32]
    L.H264E_clip_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.c_int, C.POINTER(ClipStats)]
    L.H264E_clip_rewind.argtypes = [C.c_void_p]
    L.H264E_clip_rewind.restype = None
    L.H264E_clip_read_recon.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.H264E_clip_set_ssd_output.argtypes = [C.c_void_p, C.c_void_p]
    L.H264E_clip_set_ssd_output.restype = None
    L.H264E_clip_revalidate.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.H264E_clip_restart.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int32)]
    L.H264E_clip_close.argtypes = [C.c_void_p]
    L.H264E_clip_close.restype = None
    L.h264e_hip_device_count.restype = C.c_int
    _libs[path] = L
    return L


def _err(L, what):
    return H264EError("%s: %s" % (what, (L.H264E_last_error() or b"").decode()))


class Encoder:
    """Frame-at-a-time encoder through the reference API: H264E_sizeof -> H264E_init -> H264E_encode."""

    def __init__(self, width, height, gop=20, qp=33, speed=0, kbps=0, const_input=1, vbv_size_bytes=100000 // 8, lib=None, slices=0):
        self.L = load(lib)
        self.w, self.h = width, height
        self.cp = CreateParam(width=width, height=height, gop=gop, vbv_size_bytes=vbv_size_bytes, const_input_flag=const_input,
                              enableNEON=1, num_layers=1)
        sp, ss = C.c_int(), C.c_int()
        st = self.L.H264E_sizeof(C.byref(self.cp), C.byref(sp), C.byref(ss))
        if st:
            raise H264EError("H264E_sizeof status %d" % st)
        self.sizeof_persist, self.sizeof_scratch = sp.value, ss.value
        self.persist = C.create_string_buffer(sp.value + 64)
        self.scratch = C.create_string_buffer(ss.value + 64)
        st = self.L.H264E_init(self.persist, C.byref(self.cp))
        if st:
            raise _err(self.L, "H264E_init status %d" % st)
        if slices and self.L.H264E_set_slices(self.persist, slices):
            raise H264EError("H264E_set_slices(%d) refused" % slices)
        self.rp = RunParam(encode_speed=speed)
        if kbps:
            self.rp.desired_frame_bytes = kbps * 1000 // 8 // 30  # minih264e_test.c:596-600
            self.rp.qp_min, self.rp.qp_max = 10, 50
        else:
            self.rp.qp_min = self.rp.qp_max = qp

    def encode(self, frame, frame_type=FRAME_TYPE_DEFAULT):
        """frame: uint8 array of w*h*3/2 (packed I420).  Returns the coded bytes of this frame."""
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        w, h = self.w, self.h
        base = frame.ctypes.data
        io = IoYuv((C.c_void_p * 3)(base, base + w * h, base + w * h * 5 // 4), (C.c_int * 3)(w, w // 2, w // 2))
        self.rp.frame_type = frame_type
        data, n = C.c_void_p(), C.c_int()
        st = self.L.H264E_encode(self.persist, self.scratch, C.byref(self.rp), C.byref(io), C.byref(data), C.byref(n))
        if st:
            raise _err(self.L, "H264E_encode status %d" % st)
        return C.string_at(data, n.value)

    def encode_planes(self, y, u, v, frame_type=FRAME_TYPE_DEFAULT):
        """Three separately allocated planes with any row stride (H264E_io_yuv_t, h264-lab.h:231-237): y, u, v are 2-D uint8 arrays
        (views into larger buffers are fine) whose LAST axis is contiguous; the row stride is taken from the array.  With
        const_input_flag = 0 the reconstruction is written back into these arrays (h264-lab.h:6719-6723)."""
        for a in (y, u, v):
            assert a.dtype == np.uint8 and a.ndim == 2 and a.strides[1] == 1
        io = IoYuv((C.c_void_p * 3)(y.ctypes.data, u.ctypes.data, v.ctypes.data), (C.c_int * 3)(y.strides[0], u.strides[0], v.strides[0]))
        self.rp.frame_type = frame_type
        data, n = C.c_void_p(), C.c_int()
        st = self.L.H264E_encode(self.persist, self.scratch, C.byref(self.rp), C.byref(io), C.byref(data), C.byref(n))
        if st:
            raise _err(self.L, "H264E_encode status %d" % st)
        return C.string_at(data, n.value)

    def set_vbv_state(self, vbv_size_bytes, vbv_fullness_bytes):
        """H264E_set_vbv_state (h264-lab.h:6898-6913)"""
        self.L.H264E_set_vbv_state(self.persist, vbv_size_bytes, vbv_fullness_bytes)

    def close(self):
        if self.persist is not None:
            self.L.H264E_close(self.persist)
            self.persist = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ClipEncoder:
    """Whole-clip streaming encode on one GPU (H264E_clip_* extension): consecutive frames as a temporal wavefront."""

    def __init__(self, width, height, nframes, gop=30, qp=26, speed=0, device=0, max_chains=0, lib=None,
                 clusters_in=(0, 0), idr_state=0, slices=0, kbps=0, resident=0, keep_records=0):
        self.L = load(lib)
        self.w, self.h, self.n = width, height, nframes
        self.par = ClipParam(width, height, gop, qp, speed, 100000 // 8, device, max_chains, idr_state, (C.c_int32 * 2)(*clusters_in), slices, kbps, resident, keep_records)
        self.c = C.c_void_p()
        if self.L.H264E_clip_open(C.byref(self.c), C.byref(self.par), nframes):
            raise _err(self.L, "H264E_clip_open")

    def upload(self, clip, first=0):
        clip = np.ascontiguousarray(clip, dtype=np.uint8)
        n = clip.size // (self.w * self.h * 3 // 2)
        if self.L.H264E_clip_upload(self.c, first, n, clip.ctypes.data):
            raise _err(self.L, "H264E_clip_upload")

    def generate_synth(self, first=0, nframes=None, t0=0, seed=1):
        if self.L.H264E_clip_generate_synth(self.c, first, self.n if nframes is None else nframes, t0, seed):
            raise _err(self.L, "H264E_clip_generate_synth")

    def encode(self, profile=False, rewind=True, cap=None):
        """Encode the uploaded frames that are not encoded yet (after a rewind: all of them).  Returns (bytes, sizes, stats)."""
        if rewind:
            self.L.H264E_clip_rewind(self.c)
        cap = cap or (self.w * self.h * 3 // 2 * self.n + (1 << 20))
        out = np.empty(cap, np.uint8)
        nb = C.c_size_t()
        sizes = (C.c_int * self.n)()
        st = ClipStats()
        if self.L.H264E_clip_encode(self.c, out.ctypes.data, cap, C.byref(nb), sizes, int(profile), C.byref(st)):
            raise _err(self.L, "H264E_clip_encode")
        return out[: nb.value].tobytes(), list(sizes)[: st.frames], st

    @staticmethod
    def encode_multi(encoders):
        """Encode several clips of one picture size on one device AT THE SAME TIME (H264E_clip_encode_multi: one host thread per clip,
        the clips' launches merged into one grid per round).  Returns a list of (bytes, sizes, stats), one per encoder."""
        n = len(encoders)
        L = encoders[0].L
        caps = [e.w * e.h * 3 // 2 * e.n + (1 << 20) for e in encoders]
        outs = [np.empty(c, np.uint8) for c in caps]
        sizes = [(C.c_int * e.n)() for e in encoders]
        sts = (ClipStats * n)()
        nb = (C.c_size_t * n)()
        for e in encoders:
            L.H264E_clip_rewind(e.c)
        L.H264E_clip_encode_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                              C.POINTER(C.POINTER(C.c_int)), C.POINTER(ClipStats)]
        L.H264E_clip_encode_multi.restype = C.c_int
        clips = (C.c_void_p * n)(*[e.c for e in encoders])
        outp = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        capa = (C.c_size_t * n)(*caps)
        fb = (C.POINTER(C.c_int) * n)(*[C.cast(s, C.POINTER(C.c_int)) for s in sizes])
        if L.H264E_clip_encode_multi(clips, n, outp, capa, nb, fb, sts):
            raise _err(L, "H264E_clip_encode_multi")
        return [(outs[i][: nb[i]].tobytes(), list(sizes[i])[: sts[i].frames], sts[i]) for i in range(n)]

    def revalidate(self, exact_in):
        """(restart_frame or -1, restart_state, end_state) for the exact mv_clusters state in front of this shard"""
        rf = C.c_int()
        rs, es = (C.c_int32 * 2)(), (C.c_int32 * 2)()
        if self.L.H264E_clip_revalidate(self.c, (C.c_int32 * 2)(*exact_in), C.byref(rf), rs, es):
            raise _err(self.L, "H264E_clip_revalidate")
        return rf.value, (rs[0], rs[1]), (es[0], es[1])

    def restart(self, frame, state):
        if self.L.H264E_clip_restart(self.c, frame, (C.c_int32 * 2)(*state)):
            raise _err(self.L, "H264E_clip_restart")

    def read_records(self, frame):
        """per-macroblock (mvx, mvy, type, used_cand) of an encoded frame (keep_records=1)"""
        nmb = ((self.w + 15) // 16) * ((self.h + 15) // 16)
        buf = np.empty(nmb * 2, np.int32)
        self.L.H264E_clip_read_records.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        if self.L.H264E_clip_read_records(self.c, frame, buf.ctypes.data):
            raise _err(self.L, "H264E_clip_read_records")
        mv, rest = buf[0::2], buf[1::2]
        mvx = ((mv & 0xffff) ^ 0x8000) - 0x8000
        mvy = mv >> 16
        typ = ((rest & 0xff) ^ 0x80) - 0x80
        return [(int(mvx[i]), int(mvy[i]), int(typ[i]), int((rest[i] >> 8) & 0xff)) for i in range(nmb)]

    def read_recon(self, frame):
        cw, ch = (self.w + 15) // 16 * 16, (self.h + 15) // 16 * 16
        buf = np.empty(cw * ch * 3 // 2, np.uint8)
        if self.L.H264E_clip_read_recon(self.c, frame, buf.ctypes.data):
            raise _err(self.L, "H264E_clip_read_recon")
        return buf

    def close(self):
        if self.c:
            self.L.H264E_clip_close(self.c)
            self.c = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
