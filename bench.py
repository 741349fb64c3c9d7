#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json: 1080p macroblocks/sec, achieved HBM GB/s vs peak.

A step = one pass of the encode path over one batch: a synthetic 1920x1080 I420 clip of 600 frames (synth_v1,
generated in HBM before the timed region), IPPP GOP 30, QP 26, speed 0 (BASELINE.json configs[2]).  Consecutive frames
run as a temporal wavefront inside one kernel launch (a P frame starts once its reference frame is a few macroblock
rows ahead); finished frames are exported to host-mapped memory by the kernel and validated / NAL-assembled by the
host while the launch is still running (DESIGN.md sections 4-5).  The timed region covers everything after the input
is resident: all kernel launches incl. the relaunches after a mis-speculated mv_clusters state, the export of the
coded slices, host NAL assembly into the final Annex-B stream.
With --gpus N (launched by torch.distributed.run) every rank encodes its own clip on its own GPU: clips are
independent, no data-path collective (weak scaling); torch.distributed only provides the barrier and the max-reduce.
"""
import argparse
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
BYTES_I, BYTES_P = 384 + 384, 768 + 384


def _pmc_traffic():
    """HBM bytes per h264e_mb_kernel launch from the committed rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE in
    separate runs of this same command; profiles/r01_pmc_traffic.json says how they were taken), or None"""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            return json.load(f)["bytes_per_launch"]
    except Exception:
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=FRAMES, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true", help=argparse.SUPPRESS)
    # used only by tests/test_multirank.py to run the N > 1 path on CPU: gloo instead of RCCL, the lane-loop emulation library
    # of tests/emu instead of the GPU library, a tiny picture
    ap.add_argument("--backend", default="nccl", help=argparse.SUPPRESS)
    ap.add_argument("--lib", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--size", default="%dx%d" % (W, H), help=argparse.SUPPRESS)
    a = ap.parse_args()
    w, h = (int(v) for v in a.size.split("x"))
    nmb = ((w + 15) // 16) * ((h + 15) // 16)
    on_gpu = a.backend == "nccl"

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        if on_gpu:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(a.backend)

    from __graft_entry__ import _pkg
    P = _pkg()
    frames = a.frames
    kw = {"lib": a.lib} if a.lib else {}
    enc = P.ClipEncoder(w, h, frames, gop=GOP, qp=QP, speed=0, device=local_rank if on_gpu else 0, **kw)
    enc.generate_synth(0, frames, t0=rank * frames, seed=1)   # every rank its own clip (weak scaling)

    def barrier():
        if dist is not None:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        out, sizes, st = enc.encode()
    barrier()
    t0 = time.time()
    mb_ms = splice_ms = 0.0
    launches = 0
    for _ in range(a.steps):
        out, sizes, st = enc.encode(profile=True)     # HIP events on the encoder's own stream, read after the step
        mb_ms += st.mb_kernel_ms
        splice_ms += st.splice_kernel_ms
        launches += st.kernel_launches
    barrier()
    dt = time.time() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    per_rank = None
    if dist is not None and not on_gpu:         # test hook: which stream did every rank produce?
        import hashlib
        per_rank = [None] * world
        dist.all_gather_object(per_rank, hashlib.md5(out).hexdigest())

    if rank == 0:
        total_mb = world * a.steps * frames * nmb
        value = total_mb / dt
        # useful algorithmic bytes of the clip (frames that were encoded again after a mis-speculation count once)
        alg_bytes = a.steps * nmb * sum((BYTES_P if (f % GOP) else BYTES_I) for f in range(frames))
        achieved = alg_bytes / (mb_ms * 1e-3) / 1e9 if mb_ms > 0 else 0.0
        line = {
            "metric": "1080p macroblocks/sec", "value": value, "unit": "macroblocks/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt * 1e3 / a.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "synthetic %dx%d YUV420 %d frames IPPP GOP %d QP %d on 1xMI355X per rank (BASELINE configs[2])" % (w, h, frames, GOP, QP),
                       "frames_per_step": frames, "frames_in_flight": st.chains, "fps": world * a.steps * frames / dt,
                       "coded_bytes_per_step": len(out), "relaunches_per_step": st.reencoded_gops},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": _pmc_traffic(), "kernel": "h264e_mb_kernel", "launches": launches, "avg_launch_ms": mb_ms / max(launches, 1),
                         "bytes_per_launch": alg_bytes / max(launches, 1), "splice_kernel_ms_total": splice_ms},
        }
        if per_rank is not None:
            line["config"]["per_rank_md5"] = per_rank
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
    enc.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
