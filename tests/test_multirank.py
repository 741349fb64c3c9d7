"""CPU: the N > 1 path of bench.py with world_size 2 over gloo -- every rank encodes its OWN clip (weak scaling, no
data-path collective), the barrier / max-reduce / rank-0 JSON line work, and each rank's stream is the oracle's stream
for that rank's frames.  The encode itself runs through the lane-loop emulation of the kernel sources (tests/emu), which is
what a machine without a GPU can run; the harness code under test is the one the GPU bench uses."""
import hashlib
import json
import os
import socket
import subprocess
import sys

import oracle_lib
import pkg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_two_ranks_gloo():
    w, h, frames, world = 64, 48, 4, 2
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "1",
                                       "--frames", str(frames), "--size", "%dx%d" % (w, h), "--backend", "gloo", "--lib", pkg.EMU_LIB, "--no-cpu-baseline"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1].decode()[-2000:] for o in outs]
    lines = [l for l in outs[0][0].decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][0].decode().splitlines() if l.startswith("{")], "exactly rank 0 prints the line"
    d = json.loads(lines[0])
    nmb = ((w + 15) // 16) * ((h + 15) // 16)
    assert d["n_gpus"] == world and d["scaling"] == "weak" and d["steps"] == 1 and d["unit"] == "macroblocks/s"
    assert abs(d["value"] - world * frames * nmb / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]     # whole-job aggregate over the max time
    # every rank encoded its own frames: rank r = synth_v1 frames r*frames .. (r+1)*frames-1
    want = []
    for r in range(world):
        import numpy as np
        clip = np.empty((frames, w * h * 3 // 2), np.uint8)
        for t in range(frames):
            oracle_lib.lib().synth_v1_frame(clip[t].ctypes.data, w, h, r * frames + t, 1)
        stream, _ = oracle_lib.encode_clip(clip, w, h, gop=30, qp=26)
        want.append(hashlib.md5(stream).hexdigest())
    assert d["config"]["per_rank_md5"] == want and want[0] != want[1]
