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


def make(name, w, h, n):
    if name == "synth":
        return synth.clip(w, h, n)
    return {"noise": noise, "pan": pan, "extremes": extremes}[name](w, h, n)
