#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json: 1080p macroblocks/sec, achieved HBM GB/s vs peak.

A step = one pass of the encode path over one batch: a synthetic 1920x1080 I420 clip of 600 frames (synth_v1,
generated in HBM before the timed region), IPPP GOP 30, QP 26, speed 0 (BASELINE.json configs[2]).  Consecutive frames
run as a temporal wavefront inside one kernel launch (a P frame starts once its reference frame is a few macroblock
rows ahead); finished frames are spliced, NAL-escaped and exported to host-mapped memory by the kernel, validated
(mv_clusters speculation) and appended to the Annex-B stream by the host while the launch is still running (DESIGN.md
sections 4-5).  The timed region covers everything after the input is resident: all kernel launches incl. the relaunches
after a mis-speculated mv_clusters state, the export of the coded slices, the host-side stream assembly.
The md5 of the timed output is compared with the md5 of the REFERENCE encoder's stream for the same clip
(tests/golden/golden_big.json: `parity_full_stream`).

--gpus N (launched by torch.distributed.run): by default every rank encodes its own clip on its own GPU -- clips are
independent, no data-path collective (weak scaling); torch.distributed only provides the barrier and the max-reduce.
--shard stream: the ranks encode contiguous GOP blocks of ONE stream instead (strong scaling): every rank starts from a
speculated mv_clusters state, the exact state is handed down the ranks (8 bytes per boundary through torch.distributed),
each rank re-validates and encodes again from the first GOP that consumed different start candidates
(h264-lab_amd/shard.py).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H, FRAMES, GOP, QP = 1920, 1080, 600, 30, 26
NMB = ((W + 15) // 16) * ((H + 15) // 16)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# algorithmic bytes per macroblock: input 384 B; P frames also read the co-located reference
# (384 B); every frame writes 384 B of reconstruction
READ_I, READ_P, WRITE = 384, 768, 384
PROFILE_TAG = "r04"


def _pmc_traffic():
    """HBM bytes per h264e_mb_kernel launch from the committed rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE in separate
    runs of this same command; the json says how they were taken).  STATIC: read from profiles/, not measured in this run."""
    for tag in (PROFILE_TAG, "r03", "r02", "r01"):
        try:
            with open(os.path.join(ROOT, "profiles", "%s_pmc_traffic.json" % tag)) as f:
                return json.load(f)["bytes_per_launch"], "static_from_profiles/%s_pmc_traffic.json" % tag
        except Exception:
            continue
    return None, None


def _issue_counters():
    """SQ instruction counters of the committed profile (tools/pmc_insts.sh), for the issue roofline; static like the traffic"""
    for tag in (PROFILE_TAG, "r03", "r02", "r01"):
        try:
            with open(os.path.join(ROOT, "profiles", "%s_sq_counters.json" % tag)) as f:
                return json.load(f), "static_from_profiles/%s_sq_counters.json" % tag
        except Exception:
            continue
    return None, None


def _golden_md5(w, h, frames, gop, qp):
    """md5 of the REFERENCE encoder's stream for this clip, when tests/golden/golden_big.json holds it"""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "golden_big.json")) as f:
            for g in json.load(f).values():
                if (g["w"], g["h"], g["frames"], g["flags"]) == (w, h, frames, "--qp %d --gop %d" % (qp, gop)):
                    return g["md5"]
    except Exception:
        pass
    return None


def _single_gpu_line():
    """the committed one-GPU bench line of this build (profiles/), for the N > 1 lines to show beside their own time.  STATIC."""
    for tag in (PROFILE_TAG, "r03"):
        try:
            with open(os.path.join(ROOT, "profiles", "%s_bench_line.json" % tag)) as f:
                d = json.load(f)
            return {"ms_per_step": d["ms_per_step"], "value": d["value"], "source": "static_from_profiles/%s_bench_line.json" % tag}
        except Exception:
            continue
    return None


def cpu_baseline(sample_frames=60):
    """the reference's own CPU path (oracle/_ref/encode_app_ref, gcc -O2, 1 thread) on the first frames of the same
    clip; falls back to the oracle restatement when the compiled reference is not in the snapshot"""
    import ctypes as C
    import oracle_lib
    if not os.path.exists(oracle_lib.LIB):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "all"], stdout=subprocess.DEVNULL)
    lib = oracle_lib.lib()
    with tempfile.TemporaryDirectory() as tmp:
        yuv = os.path.join(tmp, "bench_%dx%d.yuv" % (W, H))
        buf = (C.c_uint8 * (W * H * 3 // 2))()
        with open(yuv, "wb") as f:
            for t in range(sample_frames):
                lib.synth_v1_frame(buf, W, H, t, 1)
                f.write(bytes(buf))
        out = os.path.join(tmp, "o.264")
        if os.path.exists(oracle_lib.REF_APP):
            kind, cmd = "reference", [oracle_lib.REF_APP, "--input", yuv, "--output", out, "--qp", str(QP), "--gop", str(GOP)]
        else:
            kind, cmd = "port", [os.path.join(ROOT, "oracle", "build", "oracle_app"), "--input", yuv, "--output", out, "--qp", str(QP), "--gop", str(GOP)]
        t0 = time.time()
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
        dt = time.time() - t0
        ref_bytes = open(out, "rb").read()
        res = {"value": sample_frames * NMB / dt, "unit": "macroblocks/s", "cores": 1, "kind": kind,
               "sample": "first %d frames of the same 1080p synth_v1 clip, IPPP GOP %d QP %d, single thread, %.1f s" % (sample_frames, GOP, QP, dt)}
        # the same program on every host core at once: N independent processes, one GOP-aligned chunk of the clip each
        # (SURVEY.md section 8d (ii)); informational, never allowed to disturb the single-core figure above
        try:
            ncpu = max(1, min(len(os.sched_getaffinity(0)), sample_frames // GOP * 8, 16))
            chunks = []
            for k in range(ncpu):
                f = os.path.join(tmp, "chunk%d_%dx%d.yuv" % (k, W, H))
                with open(f, "wb") as fh:
                    for t in range(GOP):
                        lib.synth_v1_frame(buf, W, H, k * GOP + t, 1)
                        fh.write(bytes(buf))
                chunks.append(f)
            t0 = time.time()
            procs = [subprocess.Popen([cmd[0], "--input", f, "--output", f + ".264", "--qp", str(QP), "--gop", str(GOP)], stdout=subprocess.DEVNULL) for f in chunks]
            ok = all(p.wait() == 0 for p in procs)
            dta = time.time() - t0
            if ok:
                res["all_cores"] = {"value": ncpu * GOP * NMB / dta, "unit": "macroblocks/s", "cores": ncpu,
                                    "sample": "%d processes x one %d-frame GOP of the clip each, %.1f s" % (ncpu, GOP, dta)}
        except Exception as e:  # pragma: no cover
            res["all_cores"] = {"value": None, "sample": "failed: %r" % (e,)}
    return res, ref_bytes


def dropin_api_fps(P, frames=12):
    """the reference's per-frame API (H264E_encode: upload, one launch, result back before the next frame) on the same clip"""
    import oracle_lib
    c = oracle_lib.synth_c(W, H, frames)
    e = P.Encoder(W, H, gop=GOP, qp=QP)
    e.encode(c[0])
    t0 = time.time()
    for t in range(1, frames):
        e.encode(c[t])
    dt = time.time() - t0
    e.close()
    return (frames - 1) / dt


def e2e_pass(P, enc, frames, w, h):
    """one pass that also pays for the upload: the same clip lies in pinned host memory (fetched from the device beforehand) and
    goes over PCIe inside the timed region, in front of the encode"""
    import ctypes as C
    L = enc.L
    L.H264E_clip_host_alloc.restype = C.c_void_p
    L.H264E_clip_host_alloc.argtypes = [C.c_size_t]
    L.H264E_clip_host_free.argtypes = [C.c_void_p]
    L.H264E_clip_upload_async.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.H264E_clip_upload_wait.argtypes = [C.c_void_p]
    L.H264E_clip_download.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    fsz = w * h * 3 // 2
    p = L.H264E_clip_host_alloc(fsz * frames)
    if not p:
        return None
    if L.H264E_clip_download(enc.c, 0, frames, p):
        L.H264E_clip_host_free(p)
        return None
    t0 = time.time()
    L.H264E_clip_upload_async(enc.c, 0, frames, p)
    L.H264E_clip_upload_wait(enc.c)
    up = time.time() - t0
    out, sizes, st = enc.encode()
    dt = time.time() - t0
    L.H264E_clip_host_free(p)
    return {"fps": frames / dt, "macroblocks_per_s": frames * ((w + 15) // 16) * ((h + 15) // 16) / dt, "upload_s": up,
            "stream_md5": hashlib.md5(out).hexdigest(),
            "note": "upload of the whole clip from pinned host memory serialized in front of the encode (an upper bound on the PCIe cost: "
                    "encode_app --clip overlaps the two), then one pass"}


def multi_slice_pass(P, frames, w, h, gop, slices=8):
    """the same clip as `slices` row-band slices per frame (the reference's -DH264E_MAX_THREADS build with --threads N): slices are
    independent wavefronts and the reference restarts mv_clusters in every band, so there is next to no mis-speculation"""
    enc = P.ClipEncoder(w, h, frames, gop=gop, qp=QP, speed=0, slices=slices)
    enc.generate_synth(0, frames, t0=0, seed=1)
    enc.encode()
    t0 = time.time()
    out, sizes, st = enc.encode()
    dt = time.time() - t0
    enc.close()
    nmb = ((w + 15) // 16) * ((h + 15) // 16)
    md5 = hashlib.md5(out).hexdigest()
    want = None
    try:
        with open(os.path.join(ROOT, "tests", "golden", "golden_big.json")) as f:
            for g in json.load(f).values():
                if (g["w"], g["h"], g["frames"], g["flags"]) == (w, h, frames, "--qp %d --gop %d --threads %d" % (QP, gop, slices)):
                    want = g["md5"]
    except Exception:
        pass
    return {"slices": slices, "value": frames * nmb / dt, "unit": "macroblocks/s", "fps": frames / dt, "relaunches": st.reencoded_gops,
            "spin_relaunches": st.spin_relaunches, "processed_mb": st.processed_mbs, "discarded_mb": st.processed_mbs - st.delivered_mbs,
            "stream_md5": md5, "parity_full_stream": (md5 == want) if want else None,
            "note": "a different (multi-slice) bitstream: the reference built with -DH264E_MAX_THREADS and run with --threads %d; not the headline" % slices}


def multi_stream_pass(P, frames, w, h, gop, clips=4):
    """`clips` independent copies of the clip encoded AT THE SAME TIME on the one GPU (H264E_clip_encode_multi: the streams' launches
    merged into one grid per round, so that one stream's pipeline drains are filled by the others): aggregate rate, never the headline"""
    encs = []
    for _ in range(clips):
        e = P.ClipEncoder(w, h, frames, gop=gop, qp=QP, speed=0)
        e.generate_synth(0, frames, t0=0, seed=1)
        encs.append(e)
    P.ClipEncoder.encode_multi(encs)
    t0 = time.time()
    res = P.ClipEncoder.encode_multi(encs)
    dt = time.time() - t0
    for e in encs:
        e.close()
    nmb = ((w + 15) // 16) * ((h + 15) // 16)
    want = _golden_md5(w, h, frames, gop, QP)
    md5s = [hashlib.md5(r[0]).hexdigest() for r in res]
    return {"clips": clips, "value": clips * frames * nmb / dt, "unit": "macroblocks/s", "fps": clips * frames / dt,
            "parity_full_stream": all(m == want for m in md5s) if want else None, "relaunches": [r[2].reencoded_gops for r in res],
            "spin_relaunches": [r[2].spin_relaunches for r in res], "processed_mb": [r[2].processed_mbs for r in res],
            "discarded_mb": [r[2].processed_mbs - r[2].delivered_mbs for r in res],
            "note": "aggregate of %d single-slice streams in one launch group on ONE GPU; every stream md5-checked against the reference; not the headline" % clips}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--shard", choices=["clips", "stream"], default="clips",
                    help="clips: every rank its own clip (weak scaling, default); stream: the ranks encode GOP blocks of ONE stream (strong scaling)")
    ap.add_argument("--slices", type=int, default=0, help="row-band slices per frame (the reference's --threads build); default one slice")
    ap.add_argument("--frames", type=int, default=FRAMES, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-extras", action="store_true", help=argparse.SUPPRESS)
    # used only by tests/test_multirank.py to run the N > 1 path on CPU: gloo instead of RCCL, the lane-loop emulation library
    # of tests/emu instead of the GPU library, a tiny picture
    ap.add_argument("--backend", default="nccl", help=argparse.SUPPRESS)
    ap.add_argument("--lib", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--size", default="%dx%d" % (W, H), help=argparse.SUPPRESS)
    ap.add_argument("--gop", type=int, default=GOP, help=argparse.SUPPRESS)
    a = ap.parse_args()
    w, h = (int(v) for v in a.size.split("x"))
    nmb = ((w + 15) // 16) * ((h + 15) // 16)
    on_gpu = a.backend == "nccl"
    gop = a.gop

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        # --gpus N without the launcher's environment would silently run one rank: refuse instead
        sys.exit("bench.py --gpus %d needs %d ranks (WORLD_SIZE=%d): launch with python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." %
                 (a.gpus, a.gpus, world, a.gpus, a.gpus))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        if on_gpu:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(a.backend)
    tdev = "cuda" if on_gpu else "cpu"

    from __graft_entry__ import _pkg
    P = _pkg()
    frames = a.frames
    kw = {"lib": a.lib} if a.lib else {}
    dev = local_rank if on_gpu else 0
    stream_mode = a.shard == "stream" and world > 1
    if stream_mode:
        # ONE stream of `frames` frames: this rank's GOP block
        ranges = P.shard_ranges(frames, gop, world)
        first, end = ranges[rank] if rank < len(ranges) else (frames, frames)
        shard = P.StreamShard(w, h, first, end, gop, QP, device=dev, slices=a.slices, **kw) if end > first else None
        enc = shard.enc if shard else None
        if enc:
            enc.generate_synth(0, end - first, t0=first, seed=1)
    else:
        enc = P.ClipEncoder(w, h, frames, gop=gop, qp=QP, speed=0, device=dev, slices=a.slices, **kw)
        enc.generate_synth(0, frames, t0=rank * frames, seed=1)   # every rank its own clip (weak scaling)

    def barrier():
        if dist is not None:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    reencoded = 0

    def one_step(profile):
        nonlocal reencoded
        if not stream_mode:
            return enc.encode(profile=profile)
        st = shard.first_pass() if shard else None
        # the exact mv_clusters state goes down the ranks: rank r settles once rank r-1 is final (8 bytes per boundary)
        state = torch.zeros(2, dtype=torch.int32, device=tdev)
        for r in range(world):
            if r == rank and shard:
                exact = shard.settle((int(state[0]), int(state[1])))
                state = torch.tensor(exact, dtype=torch.int32, device=tdev)
            dist.broadcast(state, src=r)
        if shard:
            reencoded = shard.reencoded
            return shard.bytes(), [len(f) for f in shard.frames], st
        return b"", [], None

    for _ in range(a.warmup):
        out, sizes, st = one_step(False)
    barrier()
    t0 = time.time()
    mb_ms = splice_ms = 0.0
    launches = 0
    spins = relaunches_timed = processed = delivered = 0
    refill_ms = 0.0
    for _ in range(a.steps):
        out, sizes, st = one_step(True)     # HIP events on the encoder's own stream, read after the step
        if st is not None:
            mb_ms += st.mb_kernel_ms
            splice_ms += st.splice_kernel_ms
            launches += st.kernel_launches
            spins += st.spin_relaunches
            relaunches_timed += st.relaunches_timed
            refill_ms += st.first_frame_ms_after_relaunch
            processed += st.processed_mbs
            delivered += st.delivered_mbs
    barrier()
    dt = time.time() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    per_rank = None
    stream_md5 = None
    if dist is not None and (not on_gpu or stream_mode):
        # test hook / stream mode: which bytes did every rank produce?  (a few MB of coded data; outside the timed region)
        per = [None] * world
        dist.all_gather_object(per, out if stream_mode else hashlib.md5(out).hexdigest())
        if stream_mode:
            stream_md5 = hashlib.md5(b"".join(per)).hexdigest()
            per_rank = [hashlib.md5(p).hexdigest() for p in per]
            ree = [None] * world
            dist.all_gather_object(ree, reencoded)
            reencoded = ree
        else:
            per_rank = per

    if rank == 0:
        total_frames = a.steps * frames * (1 if stream_mode else world)
        total_mb = total_frames * nmb
        value = total_mb / dt
        # useful algorithmic bytes of this rank's frames (frames that were encoded again after a mis-speculation count once)
        nloc = len(sizes)
        base = (ranges[0][0] if stream_mode else 0)
        rd = a.steps * nmb * sum((READ_P if ((base + f) % gop) else READ_I) for f in range(nloc))
        wr = a.steps * nmb * nloc * WRITE
        ach_r = rd / (mb_ms * 1e-3) / 1e9 if mb_ms > 0 else 0.0
        ach_rw = (rd + wr) / (mb_ms * 1e-3) / 1e9 if mb_ms > 0 else 0.0
        traffic, traffic_src = _pmc_traffic()
        line = {
            "metric": "1080p macroblocks/sec", "value": value, "unit": "macroblocks/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt * 1e3 / a.steps, "higher_is_better": True, "scaling": "strong" if stream_mode else "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "synthetic %dx%d YUV420 %d frames IPPP GOP %d QP %d, %s (BASELINE configs[2])" %
                                   (w, h, frames, gop, QP, "ONE stream GOP-sharded over the ranks" if stream_mode else "one clip per rank on 1xMI355X each"),
                       "frames_per_step": frames, "slot_ring_frames": st.chains if st is not None else None, "fps": total_frames / dt,
                       "coded_bytes_per_step": len(out), "relaunches_per_step": (st.reencoded_gops if st is not None else None),
                       "slices_per_frame": max(a.slices, 1), "shard": a.shard,
                       # event bookkeeping of rank 0's encoder over the timed steps (H264E_clip_stats_t): launches that had to be repeated
                       # because a bounded in-kernel wait expired (a forward-progress failure a retry would hide: expected 0), what the
                       # kernel reconstructed against what was delivered, and what a stopped launch costs until frames flow again
                       "spin_relaunches_per_step": spins / a.steps,
                       "processed_mb_per_step": processed / a.steps, "discarded_mb_per_step": (processed - delivered) / a.steps,
                       "discarded_share": ((processed - delivered) / processed) if processed else None,
                       "ms_to_first_frame_after_relaunch": (refill_ms / relaunches_timed) if relaunches_timed else None,
                       "relaunches_timed": relaunches_timed},
            # achieved = algorithmic READ bytes (755.2 B/MB at GOP 30) / kernel time; read+write beside it
            "roofline": {"bound": "hbm", "achieved": ach_r, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_r / HBM_PEAK_GBS,
                         "achieved_read_write": ach_rw, "frac_read_write": ach_rw / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": "h264e_mb_kernel", "launches": launches,
                         "avg_launch_ms": mb_ms / max(launches, 1), "bytes_per_launch": rd / max(launches, 1),
                         "bytes_per_launch_read_write": (rd + wr) / max(launches, 1), "splice_kernel_ms_total": splice_ms},
        }
        ic, ic_src = _issue_counters()
        if ic:
            # what actually bounds this kernel: vector-instruction issue.  A wave64 VALU instruction occupies its SIMD for 4 cycles of
            # the counters' clock domain (MI355X_MICROARCH.md: one wave alone issues at most one per 4); 1024 SIMDs.
            line["issue_roofline"] = dict(ic, source=ic_src)
        if world > 1:
            # what the N > 1 number means: clips = every rank its own stream (throughput; the driver's scaling run);
            # stream = GOP blocks of ONE stream, where a single-slice CQP stream re-encodes most blocks (the mv_clusters state is a
            # never-forgetting counter, SURVEY F3: measured 50 / 75 / 90 % of the frames again at 2 / 4 / 8 blocks, profiles/) and
            # only row-band multi-slice streams shard exactly.  The committed one-GPU figure stands beside it.
            line["config"]["shard_mode_note"] = ("one stream per GPU, no exchange: weak scaling by construction" if not stream_mode else
                                                 "GOP blocks of ONE stream with an 8-byte mv_clusters hand-off per boundary; frames_encoded_again_per_rank says what the speculation cost")
            line["config"]["single_gpu"] = _single_gpu_line()
        if per_rank is not None:
            line["config"]["per_rank_md5"] = per_rank
        if stream_mode:
            line["config"]["stream_md5"] = stream_md5
            line["config"]["frames_encoded_again_per_rank"] = reencoded
            g = _golden_md5(w, h, frames, gop, QP) if not a.slices else None
            line["config"]["parity_full_stream"] = (stream_md5 == g) if g else None
        elif world == 1:
            g = _golden_md5(w, h, frames, gop, QP) if not a.slices else None
            line["config"]["parity_full_stream"] = (hashlib.md5(out).hexdigest() == g) if g else None
            line["config"]["stream_md5"] = hashlib.md5(out).hexdigest()
        if world == 1 and on_gpu and not a.no_extras and (w, h) == (W, H):
            try:
                line["dropin_api"] = {"fps": dropin_api_fps(P), "unit": "frames/s", "note": "1080p through H264E_encode, frame by frame (upload + one launch + result per call)"}
            except Exception as e:
                line["dropin_api"] = {"fps": None, "note": "failed: %r" % (e,)}
            try:
                line["e2e"] = e2e_pass(P, enc, frames, w, h)
            except Exception as e:
                line["e2e"] = {"fps": None, "note": "failed: %r" % (e,)}
            try:
                line["multi_slice"] = multi_slice_pass(P, frames, w, h, gop)
            except Exception as e:
                line["multi_slice"] = {"value": None, "note": "failed: %r" % (e,)}
            try:
                enc.close()             # (its slot ring is the largest allocation: give it back before four more encoders open)
                line["multi_stream"] = multi_stream_pass(P, frames, w, h, gop)
            except Exception as e:
                line["multi_stream"] = {"value": None, "note": "failed: %r" % (e,)}
        if world == 1 and not a.no_cpu_baseline:
            try:
                cb, ref_bytes = cpu_baseline()
                line["cpu_baseline"] = cb
                # parity gate: the timed stream must start with the reference's stream for the sample
                k = sum(sizes[:60])
                line["config"]["parity_vs_cpu_baseline_sample"] = bool(out[:k] == ref_bytes) if frames >= 60 else None
            except Exception as e:  # the baseline is informational: never lose the GPU line over it
                line["cpu_baseline"] = {"value": None, "unit": "macroblocks/s", "cores": 1, "kind": "reference", "sample": "failed: %r" % (e,)}
        print(json.dumps(line))
    if stream_mode:
        if shard:
            shard.close()
    else:
        enc.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
