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
    # a launch that had to be repeated because a bounded wait expired is a forward-progress failure that the retry hid: never expected
    assert st.spin_relaunches == 0, "%d launches were repeated after a bounded spin expired" % st.spin_relaunches
    if name == "bench_1080p_600":
        assert st.reencoded_gops >= 10, "the bench stream no longer exercises the relaunch path"
    if name in ("4k_240", "8k_35", "4k_1200"):
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
    assert st.spin_relaunches == 0


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
        assert st.spin_relaunches == 0


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


@pytest.mark.parametrize("w,h,members,frames", [(352, 288, 8, 24), (1280, 720, 7, 10)])
def test_full_launch_groups_of_small_pictures_on_a_fast_pan(P, w, h, members, frames):
    """as many members as a group takes at this picture size (8 at CIF, 7 at 720p), on a clip whose long vectors leave the reference
    window (far reads wait for workgroups AHEAD in the dispatch order): every member offers only a few hundred rows, which by itself would
    select the four-wave latency variant (512 resident workgroups) -- the variant is chosen from the MERGED grid instead (round-3 advisor
    finding, h264e_pool.h pick_variant); every stream is the oracle's and no launch had to be repeated"""
    import clips
    import oracle_lib
    c = clips.make("pan", w, h, frames)
    want, wsizes = oracle_lib.encode_clip(c, w, h, gop=30, qp=26)
    encs = []
    for k in range(members):
        n = frames - (k % 3)                 # members of different lengths: they leave the group at different rounds
        e = P.ClipEncoder(w, h, n, gop=30, qp=26)
        e.upload(c[:n])
        encs.append((e, n))
    res = P.ClipEncoder.encode_multi([e for e, _ in encs])
    for e, _ in encs:
        e.close()
    for (out, sizes, st), (_, n) in zip(res, encs):
        assert sizes == wsizes[:n] and out == want[: sum(wsizes[:n])]
        assert st.spin_relaunches == 0


def test_launch_group_next_to_a_plain_clip_encoder(P):
    """one thread in H264E_clip_encode_multi (three streams in one launch group), another in a plain H264E_clip_encode on the same device:
    the group's merged launch takes the device's launch token like any other persistent launch (round-3 advisor finding), so the two
    take turns instead of starving each other's workgroups; every stream is the golden one and nothing was relaunched after a spin expiry"""
    import threading
    g = GOLDEN_BIG["bench_1080p_600"]
    n = 120
    grp = []
    for k in range(3):
        ce = P.ClipEncoder(g["w"], g["h"], n, gop=30, qp=26)
        ce.generate_synth(0, n)
        grp.append(ce)
    solo = P.ClipEncoder(g["w"], g["h"], n, gop=30, qp=26)
    solo.generate_synth(0, n)
    res, errs = {}, []

    def run_group():
        try:
            res["group"] = P.ClipEncoder.encode_multi(grp)
        except Exception as ex:  # noqa: BLE001 -- reported below
            errs.append(repr(ex))

    def run_solo():
        try:
            res["solo"] = [solo.encode() for _ in range(2)]
        except Exception as ex:  # noqa: BLE001
            errs.append(repr(ex))

    th = [threading.Thread(target=run_group), threading.Thread(target=run_solo)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for ce in grp + [solo]:
        ce.close()
    assert not errs, errs
    for out, sizes, st in res["group"] + res["solo"]:
        assert sizes == g["frame_bytes"][:n] and len(out) == sum(sizes)
        assert st.spin_relaunches == 0
    assert len(set(hashlib.md5(o[0]).hexdigest() for o in res["group"] + res["solo"])) == 1
