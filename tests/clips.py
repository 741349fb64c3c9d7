"""Deterministic test clips (integer-only, no RNG library): synth_v1 plus adversarial content."""
import numpy as np

import synth


def _hash_bytes(n, salt):
    idx = np.arange(n, dtype=np.uint64) + np.uint64(salt * 0x9E3779B1 & 0xFFFFFFFF)
    return (synth._h32(idx) & np.uint64(255)).astype(np.uint8)


def noise(w, h, n, salt=1):
    """uniform random bytes in every plane: exercises intra paths, large residuals, int16 truncations"""
    return _hash_bytes(n * w * h * 3 // 2, salt).reshape(n, -1)


def pan(w, h, n, step=12):
    """smooth texture panning `step` px/frame: long motion vectors, mv_clusters dynamics"""
    big = synth.frame(w + step * n + 16, h + 16, 0)[: (w + step * n + 16) * (h + 16)].reshape(h + 16, -1)
    out = np.empty((n, w * h * 3 // 2), np.uint8)
    for t in range(n):
        out[t, : w * h] = big[4 : 4 + h, step * t : step * t + w].ravel()
        out[t, w * h :] = 128
    return out


def extremes(w, h, n, salt=7):
    """0/255 checker blocks of 4x4: saturating residuals and clipping in every stage"""
    out = np.empty((n, w * h * 3 // 2), np.uint8)
    for t in range(n):
        b = (_hash_bytes((h // 4) * (w // 4), salt + t) & 1) * 255
        out[t, : w * h] = np.kron(b.reshape(h // 4, w // 4), np.ones((4, 4), np.uint8)).ravel()
        c = (_hash_bytes(w * h // 2, salt + 100 + t) & 1) * 255
        out[t, w * h :] = c
    return out


def scene(w, h, n, salt=3):
    """what synth_v1 lacks: a static background with smooth gradients (skip macroblocks, flat intra), textured sprites that move
    at different speeds and directions with sub-sample steps and cover / uncover each other (partition shapes, intra in P
    frames at the uncovered edges), a slow global brightness drift, and a SCENE CUT in the middle (a P frame that is mostly
    intra; long vectors that mean nothing)"""
    out = np.empty((n, w * h * 3 // 2), np.uint8)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.int64)
    tex = (synth._h32((xx // 3 + 977 * (yy // 3)).astype(np.uint64) + np.uint64(salt)) & np.uint64(63)).astype(np.int64)
    for t in range(n):
        cut = t >= n // 2
        base = ((xx * (3 if cut else 1) + yy * 2) // 6) % (160 if cut else 128) + (60 if cut else 40) + t // 2        # sawtooth ramps
        y = base + ((tex >> 3) if not cut else (tex >> 4))
        for k, (sx, sy, vx4, vy4, sw, sh) in enumerate([(w // 8, h // 6, 9, 2, 48, 40), (w // 2, h // 3, -6, 5, 64, 32), (w // 3, h // 2, 3, -7, 40, 56)]):
            if cut and k == 0:
                continue
            x0 = (sx * 4 + vx4 * t) // 4 % max(w - sw, 1)
            y0 = (sy * 4 + vy4 * t) // 4 % max(h - sh, 1)
            spr = 60 + 50 * k + tex[:sh, :sw] * 2 + ((xx[:sh, :sw] + yy[:sh, :sw]) & 7) * 3
            y[y0:y0 + sh, x0:x0 + sw] = spr
        out[t, : w * h] = np.clip(y, 0, 255).astype(np.uint8).ravel()
        cw, ch = w // 2, h // 2
        u = 128 + ((xx[:ch, :cw] // 8) & 7) * (2 if cut else 1) - 4
        v = 120 + ((yy[:ch, :cw] // 8) & 7) + (t & 3)
        out[t, w * h: w * h + cw * ch] = np.clip(u, 0, 255).astype(np.uint8).ravel()
        out[t, w * h + cw * ch:] = np.clip(v, 0, 255).astype(np.uint8).ravel()
    return out


def make(name, w, h, n):
    if name == "synth":
        return synth.clip(w, h, n)
    return {"noise": noise, "pan": pan, "extremes": extremes, "scene": scene}[name](w, h, n)
