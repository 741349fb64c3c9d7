"""numpy twin of oracle/synth_v1.c: bit-identical synthetic I420 clips."""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def _h32(a):
    a = a.astype(np.uint64) & M32
    a ^= a >> np.uint64(16)
    a = (a * np.uint64(0x7FEB352D)) & M32
    a ^= a >> np.uint64(15)
    a = (a * np.uint64(0x846CA68B)) & M32
    a ^= a >> np.uint64(16)
    return a


def _lattice(ix, iy, seed):
    k = ((ix.astype(np.int64) & 0xFFFFFFFF).astype(np.uint64) * np.uint64(0x9E3779B1)) & M32
    k ^= ((iy.astype(np.int64) & 0xFFFFFFFF).astype(np.uint64) * np.uint64(0x85EBCA77)) & M32
    k ^= np.uint64(seed & 0xFFFFFFFF)
    return (_h32(k) & np.uint64(255)).astype(np.int64)


def _tex(X, Y, seed, lg):
    c = 1 << lg
    ix, iy, fx, fy = X >> lg, Y >> lg, X & (c - 1), Y & (c - 1)
    a, b = _lattice(ix, iy, seed), _lattice(ix + 1, iy, seed)
    cc, d = _lattice(ix, iy + 1, seed), _lattice(ix + 1, iy + 1, seed)
    top, bot = a * (c - fx) + b * fx, cc * (c - fx) + d * fx
    return (top * (c - fy) + bot * fy + (1 << (2 * lg - 1))) >> (2 * lg)


def frame(w, h, t, seed=1):
    OFF = 1 << 20
    y, x = np.mgrid[0:h, 0:w].astype(np.int64)
    fw, fh = max(w // 8, 32), max(h // 6, 32)
    fx0, fy0 = (w // 2 + ((10 * t) >> 2)) % (w - fw), h // 3
    bg = (_tex(4 * x + 5 * t + OFF, 4 * y + 3 * t + OFF, seed, 6) * 3 >> 2) + 32
    fg = _tex(4 * x - 10 * t + OFF, 4 * y + OFF, seed + 1, 5)
    inside = (x >= fx0) & (x < fx0 + fw) & (y >= fy0) & (y < fy0 + fh)
    k = (x.astype(np.uint64) ^ (y.astype(np.uint64) << np.uint64(12)) ^ np.uint64((t << 24) & 0xFFFFFFFF)
         ^ np.uint64((seed * 7919) & 0xFFFFFFFF)) & M32
    n = (_h32(k) % np.uint64(5)).astype(np.int64) - 2
    Y = np.clip(np.where(inside, fg, bg) + n, 0, 255).astype(np.uint8)
    yc, xc = np.mgrid[0:h // 2, 0:w // 2].astype(np.int64)
    U = (128 + ((_tex(8 * xc + 5 * t + OFF, 8 * yc + 3 * t + OFF, seed + 2, 7) - 128) >> 2)).astype(np.uint8)
    V = (128 - ((_tex(8 * xc + 5 * t + OFF, 8 * yc + 3 * t + OFF, seed + 3, 7) - 128) >> 3)).astype(np.uint8)
    return np.concatenate([Y.ravel(), U.ravel(), V.ravel()])


def clip(w, h, n, seed=1):
    return np.stack([frame(w, h, t, seed) for t in range(n)])
