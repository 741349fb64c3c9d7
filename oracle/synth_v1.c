/*
 * synth_v1 -- deterministic integer-only I420 clip generator.
 *
 * TEST INFRASTRUCTURE.  Not part of the reference: the reference ships no
 * usable test clip (sequence/foreman.* are in .MISSING_LARGE_BLOBS), so parity
 * and benchmark inputs are synthesised here.  A numpy twin lives in
 * tests/synth.py; both must produce byte-identical files (md5 pinned in
 * tests/golden/golden.json).
 *
 * usage: synth_v1 <w> <h> <nframes> <out.yuv> [seed]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "synth_v1.h"

static uint32_t h32(uint32_t a)
{
    a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
    return a;
}

static uint32_t lattice(int32_t ix, int32_t iy, uint32_t seed)
{
    return h32((uint32_t)ix*0x9E3779B1u ^ (uint32_t)iy*0x85EBCA77u ^ seed) & 255;
}

static int tex(int32_t X, int32_t Y, uint32_t seed, int lg)
{
    int32_t c = 1 << lg, ix = X >> lg, iy = Y >> lg, fx = X & (c - 1), fy = Y & (c - 1);
    int32_t a = lattice(ix, iy, seed), b = lattice(ix + 1, iy, seed);
    int32_t cc = lattice(ix, iy + 1, seed), d = lattice(ix + 1, iy + 1, seed);
    int32_t top = a*(c - fx) + b*fx, bot = cc*(c - fx) + d*fx;
    return (top*(c - fy) + bot*fy + (1 << (2*lg - 1))) >> (2*lg);
}

void synth_v1_frame(uint8_t *dst, int w, int h, int t, uint32_t seed)
{
    const int32_t OFF = 1 << 20;
    int x, y;
    int fw = w/8 > 32 ? w/8 : 32, fh = h/6 > 32 ? h/6 : 32;
    int fx0 = (w/2 + ((10*t) >> 2)) % (w - fw), fy0 = h/3;
    uint8_t *Y = dst, *U = dst + w*h, *V = U + (w/2)*(h/2);
    for (y = 0; y < h; y++)
    {
        for (x = 0; x < w; x++)
        {
            int v, n;
            if (x >= fx0 && x < fx0 + fw && y >= fy0 && y < fy0 + fh)
                v = tex(4*x - 10*t + OFF, 4*y + OFF, seed + 1, 5);
            else
                v = (tex(4*x + 5*t + OFF, 4*y + 3*t + OFF, seed, 6)*3 >> 2) + 32;
            n = (int)(h32((uint32_t)x ^ ((uint32_t)y << 12) ^ ((uint32_t)t << 24) ^ (uint32_t)(seed*7919u)) % 5) - 2;
            v += n;
            Y[y*w + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
    for (y = 0; y < h/2; y++)
    {
        for (x = 0; x < w/2; x++)
        {
            U[y*(w/2) + x] = (uint8_t)(128 + ((tex(8*x + 5*t + OFF, 8*y + 3*t + OFF, seed + 2, 7) - 128) >> 2));
            V[y*(w/2) + x] = (uint8_t)(128 - ((tex(8*x + 5*t + OFF, 8*y + 3*t + OFF, seed + 3, 7) - 128) >> 3));
        }
    }
}

#ifdef SYNTH_MAIN
int main(int argc, char **argv)
{
    int w, h, n, t;
    uint32_t seed = 1;
    uint8_t *buf;
    FILE *f;
    if (argc < 5)
    {
        fprintf(stderr, "usage: %s <w> <h> <nframes> <out.yuv> [seed]\n", argv[0]);
        return 1;
    }
    w = atoi(argv[1]); h = atoi(argv[2]); n = atoi(argv[3]);
    if (argc > 5) seed = (uint32_t)atoi(argv[5]);
    buf = (uint8_t *)malloc((size_t)w*h*3/2);
    f = fopen(argv[4], "wb");
    if (!buf || !f) return 1;
    for (t = 0; t < n; t++)
    {
        synth_v1_frame(buf, w, h, t, seed);
        fwrite(buf, (size_t)w*h*3/2, 1, f);
    }
    fclose(f);
    free(buf);
    return 0;
}
#endif
