"""Seeded random configurations for the parity sweeps (tests/test_emu_parity.py, tests/test_gpu_parity.py): picture sizes with and
without cropping (any even size >= 32), every content generator, QP 10..51, GOP 1..12, the speed levels that change decisions,
row-band slices, rate control.  Deterministic: the same list on every machine."""


def _lcg(seed):
    s = seed & 0xffffffff
    while True:
        s = (s * 1664525 + 1013904223) & 0xffffffff
        yield s >> 8


def cases(n, seed, max_pixels):
    r = _lcg(seed)
    out = []
    names = ["synth", "noise", "pan", "extremes", "scene"]
    while len(out) < n:
        w = 32 + 2 * (next(r) % 320)
        h = 32 + 2 * (next(r) % 240)
        frames = 2 + next(r) % 7
        if w * h * frames > max_pixels:
            continue
        name = names[next(r) % len(names)]
        if name == "extremes":
            w, h = (w + 3) // 4 * 4, (h + 3) // 4 * 4          # the generator tiles 4x4 blocks
        if name == "pan" and w < 64:
            continue
        if name == "synth" and (w < 48 or h < 48):
            continue                                        # synth_v1's foreground box needs room to move
        mode = next(r) % 10
        kw = dict(gop=1 + next(r) % 12, qp=10 + next(r) % 42, speed=[0, 0, 0, 1, 2, 8, 9, 10][next(r) % 8], slices=0, kbps=0)
        nmby = (h + 15) // 16
        if mode >= 7:
            kw["slices"] = 2 + next(r) % min(4, max(1, nmby - 1))
        if mode in (5, 6, 9):
            kw["kbps"] = 50 + next(r) % 800
        out.append((name, w, h, frames, kw))
    return out
