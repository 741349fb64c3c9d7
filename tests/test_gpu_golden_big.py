"""GPU: the HIP path against FULL-LENGTH streams of the reference encoder (tests/golden/golden_big.json, produced by
tests/golden/make_golden_big.py from oracle/_ref): the whole 600-frame 1080p bench stream with all its mis-speculation
relaunches, 300-frame CIF (BASELINE configs[0]/[1] stand-ins), 4K, 8K, rate control and row-band multi-slice streams."""
import hashlib
import json
import os

import pytest

import pkg

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN_BIG = json.load(open(os.path.join(HERE, "golden", "golden_big.json")))


def _flags(s):
    t = s.split()
    d = dict(zip(t[0::2], t[1::2]))
    return dict(gop=int(d.get("--gop", 20)), qp=int(d.get("--qp", 33)), speed=int(d.get("--speed", 0)), kbps=int(d.get("--kbps", 0)),
                slices=int(d.get("--threads", 0)))


@pytest.fixture(scope="module")
def P():
    p = pkg.load_pkg()
    L = p.load()
    assert L.h264e_hip_device_count() > 0, "no HIP device visible"
    return p


@pytest.mark.parametrize("name", sorted(GOLDEN_BIG))
def test_full_length_stream_matches_reference(P, name):
    """streaming clip encoder (input generated in HBM by the synth_v1 kernel, md5 of the generator pinned elsewhere)"""
    g = GOLDEN_BIG[name]
    f = _flags(g["flags"])
    ce = P.ClipEncoder(g["w"], g["h"], g["frames"], gop=f["gop"], qp=f["qp"], speed=f["speed"], slices=f["slices"], kbps=f["kbps"])
    if g.get("clip", "synth") == "synth":
        ce.generate_synth()
    else:
        import clips
        ce.upload(clips.make(g["clip"], g["w"], g["h"], g["frames"]))
    out, sizes, st = ce.encode()
    ce.close()
    assert sizes == g["frame_bytes"]
    assert len(out) == g["bytes"] and hashlib.md5(out).hexdigest() == g["md5"]
    if name == "bench_1080p_600":
        assert st.reencoded_gops >= 10, "the bench stream no longer exercises the relaunch path"
    if name in ("4k_240", "8k_35"):
        # 4K / 8K single slice at lengths that reach the abort / relaunch path, slot reuse and GOP boundaries at those geometries
        assert st.reencoded_gops >= 1, "%s no longer exercises the relaunch path" % name


@pytest.mark.parametrize("name,max_chains", [("bench_1080p_600", 64), ("bench_1080p_600_thr8", 40), ("cif_300_gop30", 16), ("1080p_30_kbps", 3)])
def test_full_length_stream_with_a_small_slot_ring(P, name, max_chains):
    """the same streams with the slot ring capped far below the clip length (frame f lives in slot f mod K, so slots are reused many
    times and every launch boundary falls somewhere else): still the reference's bytes"""
    g = GOLDEN_BIG[name]
    f = _flags(g["flags"])
    ce = P.ClipEncoder(g["w"], g["h"], g["frames"], gop=f["gop"], qp=f["qp"], speed=f["speed"], slices=f["slices"], kbps=f["kbps"], max_chains=max_chains)
    ce.generate_synth()
    out, sizes, st = ce.encode()
    ce.close()
    assert sizes == g["frame_bytes"]
    assert len(out) == g["bytes"] and hashlib.md5(out).hexdigest() == g["md5"]


def test_four_streams_in_one_launch_group(P):
    """H264E_clip_encode_multi: four copies of the 600-frame 1080p bench clip encoded at the same time on one GPU, their launches merged
    into one grid per round (include/h264e_hip.h launch groups), each with its own mis-speculation aborts and relaunches: every stream is
    the reference's"""
    g = GOLDEN_BIG["bench_1080p_600"]
    encs = []
    for k in range(4):
        ce = P.ClipEncoder(g["w"], g["h"], g["frames"], gop=30, qp=26)
        ce.generate_synth()
        encs.append(ce)
    res = P.ClipEncoder.encode_multi(encs)
    for ce in encs:
        ce.close()
    for out, sizes, st in res:
        assert sizes == g["frame_bytes"]
        assert len(out) == g["bytes"] and hashlib.md5(out).hexdigest() == g["md5"]
        assert st.reencoded_gops >= 10


def test_launch_group_of_unequal_streams(P):
    """clips of different content, length and slice count in one launch group (CIF): the members leave the group at different times
    and one of them runs a different kernel variant (all-intra: one wave per row)"""
    import clips
    import oracle_lib
    w, h = 352, 288
    specs = [("synth", 40, 10, 26, 0), ("pan", 24, 6, 28, 0), ("noise", 16, 1, 30, 0), ("synth", 30, 10, 26, 3)]
    encs, want = [], []
    for name, n, gop, qp, slices in specs:
        c = clips.make(name, w, h, n)
        want.append(oracle_lib.encode_clip(c, w, h, gop=gop, qp=qp, slices=slices))
        e = P.ClipEncoder(w, h, n, gop=gop, qp=qp, slices=slices)
        e.upload(c)
        encs.append(e)
    res = P.ClipEncoder.encode_multi(encs)
    for e in encs:
        e.close()
    for (out, sizes, st), (wb, ws) in zip(res, want):
        assert out == wb and sizes == ws


def test_launch_groups_are_batched_by_what_one_grid_can_hold(P):
    """three 4K streams: one grid carries two of that size safely (the far-read window, h264e_hip_group_join), so H264E_clip_encode_multi
    runs a group of two and then the third by itself; all three streams are the reference's"""
    g = GOLDEN_BIG["4k_30"]
    encs = []
    for k in range(3):
        ce = P.ClipEncoder(g["w"], g["h"], g["frames"], gop=30, qp=26, max_chains=32)
        ce.generate_synth()
        encs.append(ce)
    res = P.ClipEncoder.encode_multi(encs)
    for ce in encs:
        ce.close()
    for out, sizes, st in res:
        assert sizes == g["frame_bytes"] and hashlib.md5(out).hexdigest() == g["md5"]
