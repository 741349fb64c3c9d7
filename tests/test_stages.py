"""Per-stage fixtures of the REFERENCE's own functions (tests/golden/stages.json, made by oracle/stage_harness.c, which compiles
the reference header into a harness: SAD quadrants, the 16 luma quarter-sample positions, chroma bilinear, transform + dead-zone
quantiser + dequantiser + reconstruction in its four modes, one CAVLC block) against
  * the oracle's restatement of each stage (CPU)                                       -- pins the oracle below the stream level,
  * the kernel sources' wave-level stage, in the lane-loop emulation build (CPU),
  * the same wave-level stage on the GPU, through the LDS window and through the HBM path (-m gpu).
When a stream md5 differs, these say which stage."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_lib
import pkg

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "stages.json")))
QMODE = {"inter": 8, "i16": 9, "i4": 2, "chroma": 5}


def _b(h):
    return bytes.fromhex(h)


# ------------------------------------------------------------------ the oracle's restatement

def _olib():
    L = oracle_lib.lib()
    L.sad_16x16_q.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int)]
    L.interp_luma.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p]
    L.interp_chroma.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p]
    L.xform_quant.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.quant_luma_dc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.quant_chroma_dc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.recon_blocks.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_void_p, C.c_int, C.c_uint32]
    L.build_qdat.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.cavlc_block.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p]
    L.bw_init.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.bw_flush.argtypes = [C.c_void_p]
    return L


class _BitW(C.Structure):
    _fields_ = [("buf", C.c_void_p), ("cap", C.c_size_t), ("acc", C.c_uint64), ("nacc", C.c_int), ("pos", C.c_size_t)]


def _recon_flow(mode, nz, dcflag, q, recon):
    """the reconstruction exactly as mb_write / intra_choose_4x4 call it (h264-lab.h:4428-4433, 4468-4488, 4809-4811);
    q: int16 array [16][32] (qv, dq), modified in place like the reference's; recon(side, mask) does the add"""
    if mode == QMODE["inter"]:
        recon(4, (nz << 16) & 0xffffffff)
    elif mode == QMODE["i16"]:
        recon(4, 0xffff0000)
    elif mode == QMODE["i4"]:
        if nz & 1:
            recon(1, 0x80000000)
    elif dcflag or nz:
        m = nz
        if dcflag:
            for b4 in range(4):
                if ~nz & (8 >> b4):
                    q[b4, 17:32] = 0
            m = 15
        recon(2, (m << 28) & 0xffffffff)


def test_oracle_sad_quadrants():
    L = _olib()
    for c in FIX["sad"]:
        pic, blk = _b(c["pic"]), _b(c["blk"])
        s4 = (C.c_int * 4)()
        a = C.create_string_buffer(pic[c["oy"] * 64 + c["ox"]:] + bytes(64), len(pic))
        assert L.sad_16x16_q(a, 64, blk, 16, s4) == c["sad"] and list(s4) == c["sad4"]


def test_oracle_luma_and_chroma_interpolation():
    L = _olib()
    f = FIX["qpel_luma"][0]
    pic = _b(f["pic"])
    for c in f["cases"]:
        dst = C.create_string_buffer(256)
        L.interp_luma(pic, 64, 4 * c["x"] + c["dx"], 4 * c["y"] + c["dy"], c["w"], c["h"], dst)
        want = np.frombuffer(_b(c["dst"]), np.uint8).reshape(16, 16)[: c["h"], : c["w"]]
        got = np.frombuffer(dst.raw, np.uint8).reshape(16, 16)[: c["h"], : c["w"]]
        assert (got == want).all(), c
    f = FIX["qpel_chroma"][0]
    pic = _b(f["pic"])
    for c in f["cases"]:
        dst = C.create_string_buffer(256)
        L.interp_chroma(pic, 64, 8 * c["x"] + c["dx"], 8 * c["y"] + c["dy"], c["w"], c["h"], dst)
        want = np.frombuffer(_b(c["dst"]), np.uint8).reshape(16, 16)[: c["h"], : c["w"]]
        got = np.frombuffer(dst.raw, np.uint8).reshape(16, 16)[: c["h"], : c["w"]]
        assert (got == want).all(), c


def test_oracle_quantiser_tables_transform_quant_recon():
    L = _olib()
    for c in FIX["quant"]:
        mode, side = c["mode"], c["mode"] >> 1
        nblk = 1 if mode == QMODE["i4"] else side * side
        qd = np.zeros((2, 42), np.uint16)
        L.build_qdat(qd.ctypes.data, c["qp"], c["p_slice"])
        plane = 1 if mode == QMODE["chroma"] else 0
        assert qd[plane].tobytes() == _b(c["qdat"]), "quantiser table of QP %d" % c["qp"]
        q = np.zeros((16, 32), np.int16)
        dc = np.zeros(16, np.int16)
        lev = np.zeros(16, np.int16)
        inp, pred = _b(c["inp"]), _b(c["pred"])
        nz = L.xform_quant(inp, 16, pred, mode, q.ctypes.data, dc.ctypes.data, qd[plane].ctypes.data)
        dcflag = 0
        if mode == QMODE["i16"]:
            L.quant_luma_dc(q.ctypes.data, dc.ctypes.data, lev.ctypes.data, qd[plane].ctypes.data)
        if mode == QMODE["chroma"]:
            dcflag = L.quant_chroma_dc(q.ctypes.data, dc.ctypes.data, lev.ctypes.data, qd[plane].ctypes.data)
        assert (nz, dcflag) == (c["nz"], c["dcflag"]), c["mode"]
        assert q[:nblk].tobytes() == _b(c["q"]), (c["qp"], c["mode"])
        if mode & 1:
            n = 16 if mode == QMODE["i16"] else 4
            assert dc[:n].tobytes() == _b(c["dc"])[: 2 * n] and lev[:n].tobytes() == _b(c["deq_dc"])[: 2 * n]
        out = C.create_string_buffer(pred, 256)
        _recon_flow(mode, nz, dcflag, q, lambda sd, mask: L.recon_blocks(out, 16, pred, q.ctypes.data, sd, mask))
        assert out.raw == _b(c["out"]), (c["qp"], c["mode"])


def test_oracle_cavlc_block():
    L = _olib()
    for c in FIX["cavlc"]:
        buf = C.create_string_buffer(64)
        bw = _BitW()
        L.bw_init(C.byref(bw), buf, 64)
        coef = np.frombuffer(_b(c["coef"]), np.int16).copy()
        nnz = C.create_string_buffer(1)
        L.cavlc_block(C.byref(bw), coef.ctypes.data, 1 if c["maxn"] == 15 else 0, c["maxn"], c["left"] + c["top"], nnz)
        nbits = bw.pos * 8 + bw.nacc
        L.bw_flush(C.byref(bw))
        assert (nnz.raw[0], nbits) == (c["nnz"], c["nbits"]), c
        nb = (c["nbits"] + 7) // 8
        assert buf.raw[:nb] == _b(c["bits"])[:nb], c


def test_oracle_intra4_mode_choice():
    L = _olib()
    L.i4_choose.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    for c in FIX["intra4"]:
        e = _b(c["edge"])
        pred = C.create_string_buffer(64)
        sad = C.c_int()
        mode = L.i4_choose(_b(c["in"]), pred, c["avail"], e[5:13], bytes(reversed(e[0:4])), e[4], c["mpred"], c["penalty"], C.byref(sad))
        assert (mode, sad.value) == (c["mode"], c["cost"]), c
        want = np.frombuffer(_b(c["pred"]), np.uint8).reshape(4, 16)[:, :4]
        assert (np.frombuffer(pred.raw, np.uint8).reshape(4, 16)[:, :4] == want).all(), c


def _plane(h, n):
    return np.frombuffer(_b(h), np.uint8).reshape(n, n).copy()


def test_oracle_deblock_macroblock():
    L = _olib()
    L.deblock_mb.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int]
    for c in FIX["deblock"]:
        y, u, v = _plane(c["y_in"], 32), _plane(c["u_in"], 16), _plane(c["v_in"], 16)
        L.deblock_mb(y.ctypes.data + 8 * 32 + 8, 32, u.ctypes.data + 4 * 16 + 4, v.ctypes.data + 4 * 16 + 4, 16, _b(c["bs"]), c["qp"], c["qp_left"], c["qp_top"])
        assert (y == _plane(c["y_out"], 32)).all() and (u == _plane(c["u_out"], 16)).all() and (v == _plane(c["v_out"], 16)).all(), (c["mb_type"], c["qp"], c["bs"])


# ------------------------------------------------------------------ the kernel sources' stages (emulation build / GPU)

def _hook(libpath):
    P = pkg.load_pkg()
    L = P.load(libpath) if libpath else P.load()
    L.h264e_hip_pool_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.h264e_hip_selftest_stage.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_uint32, C.POINTER(C.c_int), C.c_char_p, C.c_uint32]
    L.h264e_hip_pool_destroy.argtypes = [C.c_void_p]
    pool = C.c_void_p()
    assert L.h264e_hip_pool_create(C.byref(pool), 0, 64, 48, 1, 1, 1) == 0

    def run(stage, data, args, nout):
        a = (C.c_int * 24)(*(list(args) + [0] * (24 - len(args))))
        out = C.create_string_buffer(nout)
        assert L.h264e_hip_selftest_stage(pool, stage, data, len(data), a, out, nout) == 0, P.load().h264e_hip_last_error()
        return out.raw

    return run, (lambda: L.h264e_hip_pool_destroy(pool))


def _check_stages(run):
    for window in (0, 1):
        for c in FIX["sad"]:
            r = np.frombuffer(run(1, _b(c["pic"]) + _b(c["blk"]), [c["ox"], c["oy"], window], 20), np.int32)
            assert list(r[:4]) == c["sad4"] and r[4] == c["sad"], ("sad", window)
        f = FIX["qpel_luma"][0]
        for c in f["cases"]:
            r = run(2, _b(f["pic"]), [c["x"], c["y"], c["w"], c["h"], c["dx"], c["dy"], window], 256)
            want = np.frombuffer(_b(c["dst"]), np.uint8).reshape(16, 16)[: c["h"], : c["w"]]
            assert (np.frombuffer(r, np.uint8).reshape(16, 16)[: c["h"], : c["w"]] == want).all(), ("luma", window, c["w"], c["h"], c["dx"], c["dy"])
    f = FIX["qpel_chroma"][0]
    for c in f["cases"]:
        r = np.frombuffer(run(3, _b(f["pic"]), [c["x"], c["y"], c["w"], c["h"], c["dx"], c["dy"]], 256), np.uint8).reshape(16, 16)
        want = np.frombuffer(_b(c["dst"]), np.uint8).reshape(16, 16)[: c["h"], : c["w"]]
        assert (r[: c["h"], : c["w"]] == want).all() and (r[: c["h"], 8: 8 + c["w"]] == want).all(), ("chroma", c)
    for c in FIX["quant"]:
        mode = c["mode"]
        nblk = 1 if mode == QMODE["i4"] else (mode >> 1) ** 2
        r = run(4, _b(c["inp"]) + _b(c["pred"]) + _b(c["qdat"]), [mode], 8 + 1024 + 32 + 32 + 256)
        nz, dcflag = np.frombuffer(r[:8], np.int32)
        assert (nz, dcflag) == (c["nz"], c["dcflag"]), ("quant flags", c["qp"], mode)
        assert r[8: 8 + 64 * nblk] == _b(c["q"]), ("levels / dequantised", c["qp"], mode)
        if mode & 1:
            n = 16 if mode == QMODE["i16"] else 4
            # the DC levels; (the transformed DC array itself is scratch in the kernel: its dequantised values are in q[].dq[0], compared above)
            assert r[1064: 1064 + 2 * n] == _b(c["deq_dc"])[: 2 * n], ("dc levels", c["qp"], mode)
        assert r[1096: 1096 + 256] == _b(c["out"]), ("reconstruction", c["qp"], mode)
    for c in FIX["cavlc"]:
        r = run(5, _b(c["coef"]), [1 if c["maxn"] == 15 else 0, c["maxn"], c["left"] + c["top"]], 8 + 64)
        nnz, nbits = np.frombuffer(r[:8], np.int32)
        assert (nnz, nbits) == (c["nnz"], c["nbits"]), ("cavlc", c)
        words = np.frombuffer(r[8:72], "<u4").astype(">u4").tobytes()      # the kernel's bit buffer is MSB-first 32-bit words
        nb = (c["nbits"] + 7) // 8
        assert words[:nb] == _b(c["bits"])[:nb], ("cavlc bits", c)
    for window in (0, 1):
        for ci, c in enumerate(FIX["diamond"]):
            ref = _b(FIX["diamond_refs"][c["ref"]]["pic"])
            # args[20]: which 16-lane group of the wave runs the search (it is lane-group code: every group must give the same answer)
            args = [c["px"], c["py"], c["w"], c["h"]] + c["mv_in"] + c["mv_pred"] + [c["min_sad_in"], c["qp"], c["speed"]] + c["range"] + c["limit"] + [window, ci % 4]
            r = run(8, ref + _b(c["cur"]), args, 16 + 256)
            cost, mx, my = np.frombuffer(r[:12], np.int32)
            assert (cost, [mx, my]) == (c["cost"], c["mv"]), ("motion search", window, {k: v for k, v in c.items() if k not in ("cur", "pred")})
            got = np.frombuffer(r[16:272], np.uint8).reshape(16, 16)[c["py"]: c["py"] + c["h"], c["px"]: c["px"] + c["w"]]
            want = np.frombuffer(_b(c["pred"]), np.uint8).reshape(16, 16)[: c["h"], : c["w"]]
            assert (got == want).all(), ("motion search prediction", window, c["w"], c["h"], c["mv"])
    for c in FIX["deblock"]:
        # the kernel filters on its LDS tiles: the macroblock with 4 (luma) / 2 (chroma) samples of its left and top neighbours
        yt = np.zeros((20, 24), np.uint8)
        yt[:, :20] = _plane(c["y_in"], 32)[4:24, 4:24]
        cts = []
        for k in ("u_in", "v_in"):
            t = np.zeros((10, 12), np.uint8)
            t[:, :10] = _plane(c[k], 16)[2:12, 2:12]
            cts.append(t)
        r = run(7, yt.tobytes() + cts[0].tobytes() + cts[1].tobytes() + _b(c["bs"]), [c["qp"], c["qp_left"], c["qp_top"]], 480 + 240)
        got_y = np.frombuffer(r[:480], np.uint8).reshape(20, 24)[:, :20]
        assert (got_y == _plane(c["y_out"], 32)[4:24, 4:24]).all(), ("deblock luma", c["mb_type"], c["qp"], c["bs"])
        for i, k in enumerate(("u_out", "v_out")):
            got = np.frombuffer(r[480 + 120 * i: 600 + 120 * i], np.uint8).reshape(10, 12)[:, :10]
            assert (got == _plane(c[k], 16)[2:12, 2:12]).all(), ("deblock chroma", c["mb_type"], c["qp"], c["bs"])
    for c in FIX["intra4"]:
        r = run(6, _b(c["edge"]) + bytes(3) + _b(c["in"]), [c["avail"], c["mpred"], c["penalty"]], 8 + 64)
        mode, cost = np.frombuffer(r[:8], np.int32)
        assert (mode, cost) == (c["mode"], c["cost"]), ("intra4", c)
        want = np.frombuffer(_b(c["pred"]), np.uint8).reshape(4, 16)[:, :4]
        assert (np.frombuffer(r[8:72], np.uint8).reshape(4, 16)[:, :4] == want).all(), ("intra4 prediction", c)


def test_emulated_kernel_stages_match_reference_functions():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "emu")], stdout=subprocess.DEVNULL)
    run, close = _hook(pkg.EMU_LIB)
    try:
        _check_stages(run)
    finally:
        close()


@pytest.mark.gpu
def test_gpu_kernel_stages_match_reference_functions():
    run, close = _hook(None)
    try:
        _check_stages(run)
    finally:
        close()
