/*
 * enc_kernels.h -- wave64 pixel / bit kernels of the macroblock pipeline (device code, gfx950).
 *
 * Every function is called by ALL 64 lanes of the row's wavefront (see wave.h).  LDS blocks use
 * stride 16 like the reference's scratch buffers.  "H:n" = /root/reference/src/h264-lab.h:n names
 * the reference behaviour each kernel reproduces.
 *
 * Coefficient blocks keep the reference's storage order index = 4*k_h + k_v and CAVLC scans that
 * index 15..0 -- the reference has no zig-zag; parity is defined by its bitstream.
 */
#ifndef H264E_ENC_KERNELS_H
#define H264E_ENC_KERNELS_H

#include "wave.h"
#include "tables.h"
#include "h264e_dev.h"

#define MV_NA H264E_MV_NA
#define NNZ_NA 64
#define AV_T 1
#define AV_L 2
#define AV_TL 4
#define AV_TR 8
#define QMODE_I4 2
#define QMODE_INTER 8
#define QMODE_I16 9
#define QMODE_CHROMA 5
#define QD_RND 6
#define QD_THR1 10
#define QD_THR2 18
#define MUL_LAMBDA(x, l) ((x)*(l) >> 4)

DEV int mvx(mv32 v) { return (int16_t)(v & 0xffff); }
DEV int mvy(mv32 v) { return (int16_t)((uint32_t)v >> 16); }
DEV mv32 mvmk(int x, int y) { return (mv32)(((uint32_t)y << 16) | ((uint32_t)x & 0xffff)); }
DEV mv32 mvadd(mv32 a, mv32 b) { return mvmk(mvx(a) + mvx(b), mvy(a) + mvy(b)); }
DEV mv32 mvsub(mv32 a, mv32 b) { return mvmk(mvx(a) - mvx(b), mvy(a) - mvy(b)); }
DEV mv32 mvround(mv32 a) { return mvmk((mvx(a) + 1) & ~3, (mvy(a) + 1) & ~3); }     /* H:3498 */

struct qblk_t { int16_t qv[16]; int16_t dq[16]; };

/* What a macroblock row needs of its job's task, BY VALUE in wave-uniform registers: the task itself lives in global memory, and
 * every hand-off's acquire invalidates the L1, so reading a field where it is used costs an L2 round trip on the critical path
 * of every macroblock (there are about forty such reads per macroblock). */
struct RowTask
{
    int slice_type, qp, speed, no_deblock, narrow, nslices, frame_slot;
    const uint8_t *in[3];
    int in_stride[3];
    const uint8_t *ref[3];
    uint8_t *dec[3];
    const int *dep_progress;
    mv32 clusters[2];
    const mv32 *clusters_per_mb;
};
DEV RowTask rowtask_load(const h264e_frame_task_t &T)
{
    RowTask t;
    t.slice_type = uni(T.slice_type); t.qp = uni(T.qp); t.speed = uni(T.speed); t.no_deblock = uni(T.no_deblock); t.narrow = uni(T.narrow);
    t.nslices = uni(T.nslices); t.frame_slot = uni(T.frame_slot);
    for (int c = 0; c < 3; c++) { t.in[c] = uniptr(T.in[c]); t.in_stride[c] = uni(T.in_stride[c]); t.ref[c] = uniptr(T.ref[c]); t.dec[c] = uniptr(T.dec[c]); }
    t.dep_progress = uniptr(T.dep_progress);
    t.clusters[0] = (mv32)uni(T.clusters[0]); t.clusters[1] = (mv32)uni(T.clusters[1]);
    t.clusters_per_mb = uniptr(T.clusters_per_mb);
    return t;
}

/* ------------------------------------------------------------------ row bit writer */

struct BitW
{
    uint64_t acc;       /* pending bits, right aligned */
    int nacc;           /* < 32 between calls */
    uint32_t pos;       /* words written */
    uint32_t cap;       /* capacity in words */
    int overflow;
    GLOBAL_AS uint32_t *buf;    /* global memory: MSB-first 32-bit words */
};

/* H:2688-2702: append n <= 32 bits.  Uniform: all lanes hold the same state and store the same word. */
DEV void bw_put(BitW &b, int n, uint32_t v)
{
    /* whatever the caller read from LDS (vector differences, intra modes, skip runs) is wave-uniform but a vector register to the compiler,
     * and ONE such operand makes the whole writer state a vector value for the rest of the macroblock: every later put -- the CAVLC
     * codes included -- then runs on the vector unit with 64-bit shifts (a quarter-rate instruction) and exec-mask branches */
    n = uni(n); v = (uint32_t)uni((int)v);
    b.acc = (b.acc << n) | (uint64_t)v;
    b.nacc += n;
    if (b.nacc >= 32)
    {
        b.nacc -= 32;
        /* every lane stores the same word to the same address (one memory request for the wave): a store under "lane 0 only" is a divergent
         * branch in the middle of the writer, behind which the compiler keeps the writer's state in vector registers */
        if (b.pos < b.cap) cstore32((gu8 *)(b.buf + b.pos), (uint32_t)(b.acc >> b.nacc));     /* read by the finalizer workgroup */
        else b.overflow = 1;
        b.pos++;
        b.acc &= (1ull << b.nacc) - 1;
    }
}
/* The writer's state lives in LDS between macroblocks; what comes back from LDS is a vector register as far as the compiler knows, and
 * every bw_put on it would run on the vector unit (64-bit shifts, compares and branches through the exec mask -- measured: 1.2 k of
 * the kernel's static vector instructions).  The state is wave-uniform: say so once per macroblock, and the writer runs on the scalar
 * unit. */
DEV BitW bw_uniform(const BitW &s)
{
    BitW b;
    b.acc = (uint64_t)(uint32_t)uni((int)(uint32_t)s.acc) | ((uint64_t)(uint32_t)uni((int)(uint32_t)(s.acc >> 32)) << 32);
    b.nacc = uni(s.nacc); b.pos = (uint32_t)uni((int)s.pos); b.cap = (uint32_t)uni((int)s.cap); b.overflow = uni(s.overflow);
    b.buf = uniptr(s.buf);
    return b;
}
DEV int ue_len(uint32_t v) { return 2*(32 - clz32(v + 1)) - 1; }                    /* H:3402 */
DEV void bw_ue(BitW &b, uint32_t v)                                                 /* H:2738 */
{
    int n = ue_len(v);
    if (n > 32) { bw_put(b, n - 32, 0); n = 32; }
    bw_put(b, n, v + 1);
}
DEV void bw_se(BitW &b, int v) { bw_ue(b, (uint32_t)(v > 0 ? 2*v - 1 : -2*v)); }   /* H:2760 */
DEV int se_len(int v) { return ue_len((uint32_t)(v > 0 ? 2*v - 1 : -2*v)); }        /* H:3410 */
DEV uint32_t bw_bits(const BitW &b) { return b.pos*32u + (uint32_t)b.nacc; }

/* ------------------------------------------------------------------ picture access */

struct Plane { const gu8 *p; int w, h, stride; };

/* clamped per-byte path of ref_load4: only picture borders get here, kept out of line */
NOINLINE_DEV uint32_t ref_load4_border(const gu8 *row, int w, int x)
{
    uint32_t v = 0;
    for (int k = 0; k < 4; k++)
    {
        const int xx = imin(imax(x + k, 0), w - 1);
        v |= ((cload32(row + (xx & ~3)) >> (8*(xx & 3))) & 255u) << (8*k);      /* rows start 4-byte aligned */
    }
    return v;
}

/* two consecutive aligned dwords (4-byte aligned address), coherent */
DEV uint64_t cload64x2(const gu8 *p) { return (uint64_t)cload32(p) | ((uint64_t)cload32(p + 4) << 32); }

/* four samples (x..x+3, y) from HBM, little-endian packed, coordinates clamped to the picture: this IS the
 * reference's border extension (H:2232-2248) without storing the border */
DEV uint32_t ref_load4(const Plane &P, int x, int y)
{
    y = imin(imax(y, 0), P.h - 1);
    const gu8 *r = P.p + (size_t)y*P.stride;
    if (H264E_PLAIN_UNALIGNED_LOADS && x >= 0 && x + 3 < P.w) return gload32(r + x);
    if (x >= 0 && x + 7 < P.w)
    {
        /* unaligned 4 samples = two aligned coherent dwords + byte align (x + 7 < w keeps the second dword inside the row) */
        const uint64_t d = cload64x2(r + (x & ~3));
        return alignbyte32((uint32_t)(d >> 32), (uint32_t)d, (unsigned)x & 3);
    }
    return ref_load4_border(r, P.w, x);
}

DEV uint32_t lds32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
DEV void lds32_store(uint8_t *p, uint32_t v) { memcpy(p, &v, 4); }

/* 4 bytes at an arbitrary LDS byte address: two aligned dword reads + v_alignbyte.  (The MI355X also serves ONE unaligned ds_read_b32 --
 * tests/gpu_repro/lds_unaligned.hip proves it for every alignment and width -- and that form was measured in round 4: bit-exact, 1.5 % faster
 * for a lone frame, 5-6 % SLOWER with the chip full (8 slices 29.2 -> 27.4 M MB/s): an unaligned read costs the LDS two passes, and with
 * sixteen waves per CU the LDS pipe is a shared bottleneck while the v_alignbyte goes to a vector unit with slack.  Not kept.) */
DEV uint32_t lds32u(const lu8 *p)
{
    const unsigned a = (unsigned)(uintptr_t)p & 3;
    const LDS_AS uint32_t *q = (const LDS_AS uint32_t *)(p - a);
    return alignbyte32(ld32_aligned(q + 1), ld32_aligned(q), a);
}

/*
 * Reference luma through an LDS-staged window: WIN_W x WIN_W samples around the current macroblock
 * (WIN_M samples of margin on every side), loaded once per macroblock with clamped coordinates.
 * A block whose whole footprint lies inside the window is served from LDS without per-access checks;
 * anything else (long vectors) reads HBM.
 */
#define WIN_M 24
#define WIN_W 64
#define WIN_STRIDE 68       /* 17 dwords: consecutive rows start on different LDS banks */
struct RefView
{
    Plane P; const lu8 *win; int has_win, wx0, wy0;
    const GLOBAL_AS int *dep;   /* progress counters of the frame being referenced while it is still being encoded (temporal wavefront) */
    int nmbx, nmby;
    int vw, vh;                 /* valid columns / rows of the window (h264e_dev.h: wide 64 x 64, narrow 53 x 52) */
    int *far;                   /* LDS counter of accesses that had to leave the window */
    int *fail;                  /* LDS flag: a dynamic wait of this row gave up (spin bound expired): the row must stop and report */
    const int16_t *slice_row;   /* LDS: first macroblock row of every slice of the frame, slice_row[nslices] = nmby */
    int nslices;
    unsigned spin_limit;
};

/* wave-uniform: does the sample rectangle [x0,x1] x [y0,y1] lie inside the window? */
DEV bool rv_inside(const RefView &V, int x0, int y0, int x1, int y1)
{
    return V.has_win && x0 >= V.wx0 && y0 >= V.wy0 && x1 < V.wx0 + V.vw && y1 < V.wy0 + V.vh;
}
DEV const lu8 *rv_ptr(const RefView &V, int x, int y) { return V.win + (y - V.wy0)*WIN_STRIDE + (x - V.wx0); }

/*
 * Temporal wavefront: the reference picture may still be under construction by an earlier job of the same launch.
 * The window region is covered by the static lag of the row loop; a read OUTSIDE it (long vector) first waits until
 * the producing frame has finished -- reconstructed and deblocked -- the sample rectangle [x0,x1] x [y0,y1]:
 * a sample of macroblock (X,Y) is final once row Y+1 has passed column X (its top-edge filter), which in wavefront
 * order implies row Y passed X+2; on the last row, once row Y passed X+1.  Wave-uniform.
 * Forward progress: this is the one wait that can target a workgroup LATER in the dispatch order (row Y+1 <= row+6 of the
 * previous job, since vectors are clamped to 63 samples below the macroblock row, h264-lab.h:5181-5193): at most
 * (12 - lag) dispatch keys = about (12 - lag)*(nmby+1)/2 workgroups ahead (8K: 1080), fewer than the ~2000 workgroups the
 * dispatcher keeps resident beyond the oldest unfinished one -- and if that ever fails the bound below reports it.
 */
DEV bool rv_slice_last(const RefView &V, int Y)           /* is macroblock row Y the last row of its slice (nothing below filters it)? */
{
    for (int k = 1; k <= V.nslices; k++) if (V.slice_row[k] == Y + 1) return true;
    return false;
}
/* one wait: row `drow` of the producing frame must have published `need` macroblocks */
DEV void rv_wait_row(const RefView &V, int drow, int need)
{
    const GLOBAL_AS int *flag = V.dep + drow;
    unsigned spins = 0;
    for (;;)
    {
        const int seen = uni(dep_poll(flag));       /* uni: scalar loop control */
        if (seen >= need) break;
        /* negative = the producer stopped (abort / failure); bound expired = the producer never got there.  Either way the
         * samples behind this wait are not final: flag the row, which stops, poisons its counter and (for an expiry) raises the
         * launch's error flag right after this macroblock (h264e_kernels.hip) -- nothing encoded from them is ever returned */
        if (seen < 0) { if (V.fail) *V.fail = seen; break; }
        if (++spins > V.spin_limit) { if (V.fail) *V.fail = -1; break; }
        wave_nap();
    }
}
/* sample rows y0..y1, columns up to x1.  Inside one slice the lowest macroblock row is the last to become final; row bands of
 * different slices advance independently, so every slice the rectangle touches is waited for at its lowest row inside it. */
DEV void rv_wait_rect(const RefView &V, int y0, int x1, int y1)
{
    if (V.far) *V.far += 1;
    if (!V.dep) return;
    const int X = imin(imax(x1, 0), V.P.w - 1) >> 4, Ya = imin(imax(y0, 0), V.P.h - 1) >> 4, Yb = imin(imax(y1, 0), V.P.h - 1) >> 4;
    const int need = imin(X + 2, V.nmbx);
    for (int Y = Ya; Y <= Yb; Y++)
    {
        const bool last = rv_slice_last(V, Y);
        if (last) rv_wait_row(V, Y, need);
        else if (Y == Yb) rv_wait_row(V, Y + 1, need);
    }
    consumer_acquire();
}

/* The same wait inside a lane-group section (wave.h): rectangle and counters differ from group to group, so nothing here goes
 * through a scalar register; every lane polls until ITS group's rows are there (the loop ends when all lanes have left it). */
DEV void rv_wait_rect_g(const RefView &V, int y0, int x1, int y1)
{
    if (V.far) grp_count(V.far);
    if (!V.dep) return;
    const int X = imin(imax(x1, 0), V.P.w - 1) >> 4, Ya = imin(imax(y0, 0), V.P.h - 1) >> 4, Yb = imin(imax(y1, 0), V.P.h - 1) >> 4;
    const int need = imin(X + 2, V.nmbx);
    for (int Y = Ya; Y <= Yb; Y++)
    {
        const bool last = rv_slice_last(V, Y);
        if (!last && Y != Yb) continue;
        const GLOBAL_AS int *flag = V.dep + (last ? Y : Y + 1);
        unsigned spins = 0;
        for (;;)
        {
            const int seen = dep_poll(flag);
            if (seen >= need) break;
            if (seen < 0) { if (V.fail) *V.fail = seen; break; }
            if (++spins > V.spin_limit) { if (V.fail) *V.fail = -1; break; }
            wave_nap();
        }
    }
    consumer_acquire();
}

/* The window from HBM: four lanes per window row, 16 bytes each -- one wave instruction reads sixteen rows as whole 64-byte segments (four
 * instructions for the window; one lane per row and 8 bytes per load were seven instructions touching 52 different lines each) -- when the
 * window's columns lie inside the picture (uniform test); clamped dword loads, one lane per row, at the picture's left / right border. */
DEV void wave_load_window(uint8_t *win, const Plane &P, int wx0, int wy0, int narrow)
{
    /* narrow geometry (h264e_dev.h): only 53 columns x 52 rows of the window are ever read */
    const int rows = narrow ? H264E_NARROW_VH : WIN_W;
    const bool interior = wx0 >= 0 && wx0 + WIN_W <= P.w;
    if (interior)
    {
        u32x4 v[4];
        WAVE_FOR(l)
        {
            const int seg = l & 3;
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const int r = (l >> 2) + 16*k;
                if (r < rows)
                {
                    const int y = imin(imax(wy0 + r, 0), P.h - 1);
                    v[k] = cload128(P.p + (size_t)y*P.stride + wx0 + 16*seg);           /* wx0 is a multiple of 8, rows of 16: dword aligned, the 16 bytes inside one 64-byte segment */
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const int r = (l >> 2) + 16*k;
                if (r < rows)
                {
                    uint8_t *d = win + r*WIN_STRIDE + 16*seg;
                    lds32_store(d, v[k].x); lds32_store(d + 4, v[k].y); lds32_store(d + 8, v[k].z); lds32_store(d + 12, v[k].w);
                }
            }
        }
    } else
    {
        const int nq = narrow ? (H264E_NARROW_VW + 7)/8 : WIN_W/8;
        WAVE_FOR(l)
        {
            if (l < rows)
            {
                const int y = imin(imax(wy0 + l, 0), P.h - 1);
                for (int g = 0; g < 2*nq; g++) lds32_store(win + l*WIN_STRIDE + 4*g, ref_load4(P, wx0 + 4*g, y));
            }
        }
    }
    wave_sync();
}

/* ------------------------------------------------------------------ SAD (H:2162-2192) */

/* SAD of the w x h block at integer position (x0,y0) of R against LDS block b (stride 16).  4 samples per lane. */
DEV int wave_sad_ref(const RefView &R, int x0, int y0, const uint8_t *b, int w, int h)
{
    const int g = w >> 2, n = g*h;
    if (rv_inside(R, x0, y0, x0 + w - 1, y0 + h - 1))
    {
        const lu8 *base = rv_ptr(R, x0, y0);
        return wave_sum([&](int l) -> int {
            if (l >= n) return 0;
            int r = l >> (g >> 1), c = l & (g - 1);              /* g is 2 or 4 */
            return (int)sad4_u8(lds32u(base + r*WIN_STRIDE + 4*c), lds32(b + 16*r + 4*c), 0);
        });
    }
    rv_wait_rect(R, y0, x0 + w - 1, y0 + h - 1);
    return wave_sum([&](int l) -> int {
        if (l >= n) return 0;
        int r = l >> (g >> 1), c = l & (g - 1);              /* g is 2 or 4 */
        return (int)sad4_u8(ref_load4(R.P, x0 + 4*c, y0 + r), lds32(b + 16*r + 4*c), 0);
    });
}

/* the same for a lane group (wave.h): the block's w*h/4 dwords are dealt to the group's 16 lanes, 16 per pass */
DEV int grp_sad_ref(const RefView &R, int x0, int y0, const uint8_t *b, int w, int h)
{
    const int g = w >> 2, npass = (g*h) >> 4;
    if (rv_inside(R, x0, y0, x0 + w - 1, y0 + h - 1))
    {
        const lu8 *base = rv_ptr(R, x0, y0);
        return grp_sum([&](int i) -> int {
            uint32_t s = 0;
            for (int k = 0; k < npass; k++)
            {
                const int d = i + 16*k, r = d >> (g >> 1), c = d & (g - 1);              /* g is 2 or 4 */
                s = sad4_u8(lds32u(base + r*WIN_STRIDE + 4*c), lds32(b + 16*r + 4*c), s);
            }
            return (int)s;
        });
    }
    rv_wait_rect_g(R, y0, x0 + w - 1, y0 + h - 1);
    return grp_sum([&](int i) -> int {
        uint32_t s = 0;
        for (int k = 0; k < npass; k++)
        {
            const int d = i + 16*k, r = d >> (g >> 1), c = d & (g - 1);
            s = sad4_u8(ref_load4(R.P, x0 + 4*c, y0 + r), lds32(b + 16*r + 4*c), s);
        }
        return (int)s;
    });
}

/* the same 16x16 SAD with its 8x8 quadrant sums for a lane group (wave.h): four groups take four candidate positions at once */
DEV int grp_sad_ref_q(const RefView &R, int x0, int y0, const uint8_t *b, int sad4[4])
{
    if (rv_inside(R, x0, y0, x0 + 15, y0 + 15))
    {
        const lu8 *base = rv_ptr(R, x0, y0);
        grp_sum4([&](int i, int *v) {
            uint32_t top = 0, bot = 0;
            const int c = i & 3;
            for (int k = 0; k < 4; k++)
            {
                const int r = (i >> 2) + 4*k;
                const uint32_t s = sad4_u8(lds32u(base + r*WIN_STRIDE + 4*c), lds32(b + 16*r + 4*c), 0);
                if (k < 2) top += s; else bot += s;
            }
            v[c >> 1] = (int)top; v[2 + (c >> 1)] = (int)bot;
        }, sad4);
    } else
    {
        rv_wait_rect_g(R, y0, x0 + 15, y0 + 15);
        grp_sum4([&](int i, int *v) {
            uint32_t top = 0, bot = 0;
            const int c = i & 3;
            for (int k = 0; k < 4; k++)
            {
                const int r = (i >> 2) + 4*k;
                const uint32_t s = sad4_u8(ref_load4(R.P, x0 + 4*c, y0 + r), lds32(b + 16*r + 4*c), 0);
                if (k < 2) top += s; else bot += s;
            }
            v[c >> 1] = (int)top; v[2 + (c >> 1)] = (int)bot;
        }, sad4);
    }
    return sad4[0] + sad4[1] + sad4[2] + sad4[3];
}

/* 16x16 SAD with the four 8x8 quadrant sums (H:2178-2187) */
DEV int wave_sad_ref_q(const RefView &R, int x0, int y0, const uint8_t *b, int sad4[4])
{
    if (rv_inside(R, x0, y0, x0 + 15, y0 + 15))
    {
        const lu8 *base = rv_ptr(R, x0, y0);
        wave_sum4([&](int l, int *v) {
            int r = l >> 2, c = l & 3;
            v[(r >> 3)*2 + (c >> 1)] = (int)sad4_u8(lds32u(base + r*WIN_STRIDE + 4*c), lds32(b + 16*r + 4*c), 0);
        }, sad4);
    } else
    {
        rv_wait_rect(R, y0, x0 + 15, y0 + 15);
        wave_sum4([&](int l, int *v) {
            int r = l >> 2, c = l & 3;
            v[(r >> 3)*2 + (c >> 1)] = (int)sad4_u8(ref_load4(R.P, x0 + 4*c, y0 + r), lds32(b + 16*r + 4*c), 0);
        }, sad4);
    }
    return sad4[0] + sad4[1] + sad4[2] + sad4[3];
}

DEV int wave_sad_lds(const uint8_t *a, const uint8_t *b, int w, int h)
{
    const int g = w >> 2, n = g*h;
    return wave_sum([&](int l) -> int {
        if (l >= n) return 0;
        int r = l >> (g >> 1), c = l & (g - 1);              /* g is 2 or 4 */
        return (int)sad4_u8(lds32(a + 16*r + 4*c), lds32(b + 16*r + 4*c), 0);
    });
}

DEV int wave_sad_lds_q(const uint8_t *a, const uint8_t *b, int sad4[4])
{
    wave_sum4([&](int l, int *v) {
        int r = l >> 2, c = l & 3;
        v[(r >> 3)*2 + (c >> 1)] = (int)sad4_u8(lds32(a + 16*r + 4*c), lds32(b + 16*r + 4*c), 0);
    }, sad4);
    return sad4[0] + sad4[1] + sad4[2] + sad4[3];
}

/* ------------------------------------------------------------------ inter prediction */

DEV int tap6(int a, int b, int c, int d, int e, int f) { return a - 5*b + 20*c + 20*d - 5*e + f; }
/* the same filter on SECOND-stage operands (15-bit sums of the first stage): (a + f) + 5*(4*(c + d) - (b + e)) in shifts and adds -- the compiler
 * cannot know that they fit 24 bits and would take 32-bit multiplies, which run at a quarter of the rate of an add on this machine (for byte
 * operands it finds 24-bit multiply-adds with byte selects by itself: measured, the shift form is slower there) */
DEV int tap6w(int a, int b, int c, int d, int e, int f)
{
    const int t = (c + d)*4 - (b + e);           /* (x*4: a shift / shift-add for the compiler, and defined for negative x) */
    return (a + f) + t*4 + t;
}

/*
 * Standard H.264 quarter-sample luma interpolation (H:2079-2131) of 4 adjacent samples, fraction (fx,fy): half
 * samples from the 6-tap filter, quarter samples as rounded averages of the two nearest integer/half samples.
 * ld(dx, dy) returns the 4 reference samples at (x + dx .. x + dx + 3, y + dy).  Everything is unrolled over
 * compile-time indices so the tap arrays live in registers; the position dispatch is wave-uniform.
 */
template <class LD> DEV uint32_t interp_core(LD ld, int fx, int fy)
{
    if (!(fx | fy)) return ld(0, 0);
    const int pos = fx + 4*fy;
    int A[4], B[4];
    bool avg = true;
    if (fy == 0)
    {
        const uint32_t a = ld(-4, 0), b = ld(0, 0), c = ld(4, 0);
        int p[12];
#pragma unroll
        for (int k = 0; k < 4; k++) { p[k] = (int)((a >> (8*k)) & 255); p[4 + k] = (int)((b >> (8*k)) & 255); p[8 + k] = (int)((c >> (8*k)) & 255); }
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            B[i] = clip255((tap6(p[i + 2], p[i + 3], p[i + 4], p[i + 5], p[i + 6], p[i + 7]) + 16) >> 5);
            A[i] = pos == 3 ? p[5 + i] : p[4 + i];
        }
        if (pos == 2) { avg = false; A[0] = B[0]; A[1] = B[1]; A[2] = B[2]; A[3] = B[3]; }
    } else if (fx == 0)
    {
        /* vertical only: d h n */
        int cb[6][4];
#pragma unroll
        for (int r = 0; r < 6; r++)
        {
            const uint32_t b = ld(0, r - 2);
#pragma unroll
            for (int k = 0; k < 4; k++) cb[r][k] = (int)((b >> (8*k)) & 255);
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            A[i] = clip255((tap6(cb[0][i], cb[1][i], cb[2][i], cb[3][i], cb[4][i], cb[5][i]) + 16) >> 5);
            B[i] = fy == 3 ? cb[3][i] : cb[2][i];
        }
        if (fy == 2) avg = false;
    } else
    {
        int th[6][4], cb[6][5];
#pragma unroll
        for (int r = 0; r < 6; r++)
        {
            const uint32_t a = ld(-4, r - 2), b = ld(0, r - 2), c = ld(4, r - 2);
            int p[12];
#pragma unroll
            for (int k = 0; k < 4; k++) { p[k] = (int)((a >> (8*k)) & 255); p[4 + k] = (int)((b >> (8*k)) & 255); p[8 + k] = (int)((c >> (8*k)) & 255); }
#pragma unroll
            for (int i = 0; i < 4; i++) th[r][i] = tap6(p[i + 2], p[i + 3], p[i + 4], p[i + 5], p[i + 6], p[i + 7]);
#pragma unroll
            for (int i = 0; i < 5; i++) cb[r][i] = p[4 + i];
        }
        const int vd = (fx == 3) ? 1 : 0;   /* vertical half sample m (column x+1) instead of h */
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const int hv = vd ? clip255((tap6(cb[0][i + 1], cb[1][i + 1], cb[2][i + 1], cb[3][i + 1], cb[4][i + 1], cb[5][i + 1]) + 16) >> 5)
                              : clip255((tap6(cb[0][i], cb[1][i], cb[2][i], cb[3][i], cb[4][i], cb[5][i]) + 16) >> 5);
            const int hd = clip255((tap6w(th[0][i], th[1][i], th[2][i], th[3][i], th[4][i], th[5][i]) + 512) >> 10);
            const int hh = clip255(((fy == 3 ? th[3][i] : th[2][i]) + 16) >> 5);    /* b (row y) or s (row y+1) */
            if (fx == 2)      { A[i] = hd; B[i] = hh; }                        /* f j q */
            else if (fy == 2) { A[i] = hv; B[i] = hd; }                        /* i k */
            else              { A[i] = hh; B[i] = hv; }                        /* e g p r */
        }
        if (fy == 2 && fx == 2) avg = false;                                   /* j */
    }
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) out |= (uint32_t)(avg ? (A[i] + B[i] + 1) >> 1 : A[i]) << (8*i);
    return out;
}

/* the two instantiations, out of line (one copy each in the kernel): from the LDS window, or from HBM with clamping */
NOINLINE_DEV uint32_t interp4_win(const lu8 *at, int fx, int fy)
{
    return interp_core([&](int dx, int dy) -> uint32_t { return lds32u(at + dy*WIN_STRIDE + dx); }, fx, fy);
}
NOINLINE_DEV uint32_t interp4_hbm(const gu8 *p, int w, int h, int stride, int x, int y, int fx, int fy)
{
    Plane P; P.p = p; P.w = w; P.h = h; P.stride = stride;
    return interp_core([&](int dx, int dy) -> uint32_t { return ref_load4(P, x + dx, y + dy); }, fx, fy);
}

/*
 * The three half-sample neighbours the sub-pel search needs around a full-pel position, in one pass over the window:
 * at = window pointer at integer sample (ix, iy), the top-left of the 2x2 integer cell that contains all of them;
 * (ox, oy) in {0,1}^2 = offset of the full-pel position (x, y) inside that cell.  Returns {full-pel samples at (x,y),
 * horizontal half sample on row y, vertical half sample in column x, centre half sample}, 4 samples each.  Same
 * arithmetic as interp_core for (2,0), (0,2) and (2,2): six rows are loaded once, their six horizontal filters serve the
 * centre sample and (row y) the horizontal one, the vertical filter runs on the integer columns of the same rows.
 */
struct hp4_t { uint32_t x, y, z, w; };
DEV int shr_opaque(int v, int s)
{
    return opaque_int(v >> s);
}
/* One window row for the half-sample filters: the twelve samples at q - 4 .. q + 7 from FOUR aligned dwords (a lane's three overlapping
 * unaligned dwords are six aligned reads otherwise: a third fewer LDS bytes, and the LDS pipe is the shared resource of a full chip) */
DEV void hp_row_load(const lu8 *q, uint32_t &a, uint32_t &b, uint32_t &c)
{
    const unsigned sh = (unsigned)(uintptr_t)q & 3;
    const LDS_AS uint32_t *w = (const LDS_AS uint32_t *)(q - 4 - sh);
    const uint32_t d0 = ld32_aligned(w), d1 = ld32_aligned(w + 1), d2 = ld32_aligned(w + 2), d3 = ld32_aligned(w + 3);
    a = alignbyte32(d1, d0, sh); b = alignbyte32(d2, d1, sh); c = alignbyte32(d3, d2, sh);
}

/*
 * The same three half-sample planes for `nrows` (1, 2 or 4) CONSECUTIVE rows of one 4-sample column group: row k needs the horizontal
 * filters of window rows k-2 .. k+3, so consecutive rows share five of their six -- nrows + 5 filtered rows instead of 6 * nrows (a lane
 * of the sub-pel search owns consecutive rows for exactly this reason: 9 instead of 24 for a 16x16 partition).  The filtered rows
 * rotate through six register sets (everything is unrolled: the indices are compile-time); emit(k, planes) is called for row k.
 * at = window pointer at integer sample (ix, iy) of row 0 = the top-left of the 2x2 integer cell that contains the three positions;
 * (ox, oy) in {0,1}^2 = offset of the full-pel position inside that cell.  emit gets {full-pel samples, horizontal half sample on the row,
 * vertical half sample in the column, centre half sample}, 4 samples each: same arithmetic as interp_core for (2,0), (0,2) and (2,2).
 */
template <class EMIT> DEV void halfpel3_rows(const lu8 *at, int ox, int oy, int nrows, EMIT emit)
{
    int th[6][4];
    uint32_t cx[6];
#pragma unroll
    for (int j = 0; j < 9; j++)
    {
        if (j < nrows + 5)
        {
            const int s = j % 6;
            uint32_t a, b, c;
            hp_row_load(at + (j - 2)*WIN_STRIDE, a, b, c);
            int p[12];
#pragma unroll
            for (int k = 0; k < 4; k++) { p[k] = (int)((a >> (8*k)) & 255); p[4 + k] = (int)((b >> (8*k)) & 255); p[8 + k] = (int)((c >> (8*k)) & 255); }
#pragma unroll
            for (int i = 0; i < 4; i++) th[s][i] = tap6(p[i + 2], p[i + 3], p[i + 4], p[i + 5], p[i + 6], p[i + 7]);
            cx[s] = ox ? alignbyte32(c, b, 1) : b;                 /* the integer columns the vertical filter runs on */
            if (j >= 5)
            {
                /* rows j-5 .. j are in the sets: output row j - 5 */
                const int r0 = (j - 5) % 6, r1 = (j - 4) % 6, r2 = (j - 3) % 6, r3 = (j - 2) % 6, r4 = (j - 1) % 6, r5 = j % 6;
                hp4_t o;
                o.x = oy ? cx[r3] : cx[r2];
                o.y = o.z = o.w = 0;
#pragma unroll
                for (int i = 0; i < 4; i++)
                {
#define CXB(r) ((int)((cx[r] >> (8*i)) & 255))
                    /* shr_opaque: keeps the compiler from fusing "shift, clamp, pack" of two results into v_ashr_pk_u8_i32, whose
                     * results did not match the separate instructions on gfx950 (ROCm 7.2) in this function (tests/gpu_repro/ashr_pk.hip) */
                    const int fh = clip255(shr_opaque((oy ? th[r3][i] : th[r2][i]) + 16, 5));
                    const int fv = clip255(shr_opaque(tap6(CXB(r0), CXB(r1), CXB(r2), CXB(r3), CXB(r4), CXB(r5)) + 16, 5));
                    const int fd = clip255(shr_opaque(tap6w(th[r0][i], th[r1][i], th[r2][i], th[r3][i], th[r4][i], th[r5][i]) + 512, 10));
#undef CXB
                    o.y |= (uint32_t)fh << (8*i); o.z |= (uint32_t)fv << (8*i); o.w |= (uint32_t)fd << (8*i);
                }
                emit(j - 5, o);
            }
        }
    }
}

/* 4 interpolated samples at integer position (x,y); `inside` (wave-uniform) says the block's footprint is in the window */
DEV uint32_t interp_luma4(const RefView &R, bool inside, int x, int y, int fx, int fy)
{
    if (inside) return interp4_win(rv_ptr(R, x, y), fx, fy);
    return interp4_hbm(R.P.p, R.P.w, R.P.h, R.P.stride, x, y, fx, fy);
}
/* footprint of a w x h block at integer position (ix,iy) for any fraction: 6-tap support plus the dword loads' slack */
DEV bool rv_inside_interp(const RefView &R, int ix, int iy, int w, int h)
{
    return rv_inside(R, ix - 4, iy - 2, ix + w + 3, iy + h + 2);
}

/* H:4905-4910 interpolate_luma: w x h block whose top-left is (bx,by) + mv (absolute quarter-pel) -> LDS dst */
DEV void wave_interp_luma(const RefView &R, int bx, int by, mv32 mv, int w, int h, uint8_t *dst)
{
    const int g = w >> 2, n = g*h, ix = bx + (mvx(mv) >> 2), iy = by + (mvy(mv) >> 2), fx = mvx(mv) & 3, fy = mvy(mv) & 3;
    const bool inside = rv_inside_interp(R, ix, iy, w, h);
    if (!inside) rv_wait_rect(R, iy - 2, ix + w + 3, iy + h + 2);
    WAVE_FOR(l)
    {
        if (l < n)
        {
            int r = l >> (g >> 1), c = l & (g - 1);              /* g is 2 or 4 */
            lds32_store(dst + 16*r + 4*c, interp_luma4(R, inside, ix + 4*c, iy + r, fx, fy));
        }
    }
    wave_sync();
}

/* the same for a lane group */
DEV void grp_interp_luma(const RefView &R, int bx, int by, mv32 mv, int w, int h, uint8_t *dst)
{
    const int g = w >> 2, npass = (g*h) >> 4, ix = bx + (mvx(mv) >> 2), iy = by + (mvy(mv) >> 2), fx = mvx(mv) & 3, fy = mvy(mv) & 3;
    const bool inside = rv_inside_interp(R, ix, iy, w, h);
    if (!inside) rv_wait_rect_g(R, iy - 2, ix + w + 3, iy + h + 2);
    GRP_FOR(i)
    {
        for (int k = 0; k < npass; k++)
        {
            const int d = i + 16*k, r = d >> (g >> 1), c = d & (g - 1);
            lds32_store(dst + 16*r + 4*c, interp_luma4(R, inside, ix + 4*c, iy + r, fx, fy));
        }
    }
    wave_sync();
}

/* H:2065-2077 rounded average of two LDS blocks */
DEV void wave_avg(const uint8_t *a, const uint8_t *b, uint8_t *d, int w, int h)
{
    const int g = w >> 2, n = g*h;
    WAVE_FOR(l)
    {
        if (l < n)
        {
            int r = l >> (g >> 1), c = l & (g - 1);              /* g is 2 or 4 */
            uint32_t x = lds32(a + 16*r + 4*c), y = lds32(b + 16*r + 4*c), o = 0;
            for (int k = 0; k < 4; k++) o |= ((((x >> (8*k)) & 255) + ((y >> (8*k)) & 255) + 1) >> 1) << (8*k);
            lds32_store(d + 16*r + 4*c, o);
        }
    }
    wave_sync();
}

DEV void wave_copy_wh(uint8_t *d, const uint8_t *s, int w, int h)
{
    const int g = w >> 2, n = g*h;
    WAVE_FOR(l)
    {
        if (l < n)
        {
            int r = l >> (g >> 1), c = l & (g - 1);              /* g is 2 or 4 */
            lds32_store(d + 16*r + 4*c, lds32(s + 16*r + 4*c));
        }
    }
    wave_sync();
}

/*
 * H:2133-2160 + H:4915-4947: 1/8-sample bilinear chroma prediction of one partition for both planes.
 * (cx,cy) = partition position in the chroma plane, mv = LUMA vector (absolute quarter-pel), dst = U at
 * column 0, V at column 8 (stride 16), already offset to the partition.
 */
DEV void wave_interp_chroma(const RefView &RL, const Plane &RU, const Plane &RV, int cx, int cy, mv32 mv, int w, int h, uint8_t *dst)
{
    {   /* luma-sample rectangle this chroma block corresponds to: inside the window it is covered by the row loop's lag */
        const int lx0 = 2*(cx + (mvx(mv) >> 3)), ly0 = 2*(cy + (mvy(mv) >> 3));
        if (!rv_inside(RL, lx0, ly0, lx0 + 2*w + 3, ly0 + 2*h + 3)) rv_wait_rect(RL, ly0, lx0 + 2*w + 3, ly0 + 2*h + 3);
    }
    const int g = w >> 2, n = g*h, dx = mvx(mv) & 7, dy = mvy(mv) & 7;
    const int ix = cx + (mvx(mv) >> 3), iy = cy + (mvy(mv) >> 3);
    const int A = (8 - dx)*(8 - dy), B = dx*(8 - dy), C = (8 - dx)*dy, D = dx*dy;
    WAVE_FOR(l)
    {
        int pl = l >= 32, k = l & 31;
        if (k < n)
        {
            const Plane &R = pl ? RV : RU;
            int r = k/g, c = k - r*g, p0[5], p1[5];
            uint32_t a = ref_load4(R, ix + 4*c, iy + r), b = ref_load4(R, ix + 4*c + 4, iy + r);
            uint32_t e = ref_load4(R, ix + 4*c, iy + r + 1), f = ref_load4(R, ix + 4*c + 4, iy + r + 1), o = 0;
            for (int i = 0; i < 4; i++) { p0[i] = (int)((a >> (8*i)) & 255); p1[i] = (int)((e >> (8*i)) & 255); }
            p0[4] = (int)(b & 255); p1[4] = (int)(f & 255);
            for (int i = 0; i < 4; i++)
            {
                int v = (dx | dy) ? (A*p0[i] + B*p0[i + 1] + C*p1[i] + D*p1[i + 1] + 32) >> 6 : p0[i];
                o |= (uint32_t)v << (8*i);
            }
            lds32_store(dst + 8*pl + 16*r + 4*c, o);
        }
    }
    wave_sync();
}

/* ------------------------------------------------------------------ intra prediction */

/* H:1625-1651: mean of the available edges of n = 1 << lg samples each, 128 when none */
DEV int dc_pred(const uint8_t *left, int have_left, const uint8_t *top, int have_top, int lg)
{
    const int n = 1 << lg;
    int s = 0, sh = lg - 1;
    if (have_left) { for (int i = 0; i < n; i++) s += left[i]; sh++; }
    if (have_top)  { for (int i = 0; i < n; i++) s += top[i];  sh++; }
    if (sh < lg) return 128;
    return (s + (1 << (sh - 1))) >> sh;
}

/* H:1677-1714: 16x16 luma prediction, mode 0 V / 1 H / 2 DC, into LDS dst; left/top are LDS lines, avail says which exist */
DEV void wave_pred16(uint8_t *dst, const uint8_t *left, const uint8_t *top, int avail, int mode)
{
    const int dc = mode == 2 ? dc_pred(left, avail & AV_L, top, avail & AV_T, 4) : 0;
    WAVE_FOR(l)
    {
        int r = l >> 2, c = l & 3;
        uint32_t v = mode == 0 ? lds32(top + 4*c) : (uint32_t)(mode == 1 ? left[r] : dc)*0x01010101u;
        lds32_store(dst + 16*r + 4*c, v);
    }
    wave_sync();
}

/* H:1716-1781: 8x8 U | V prediction (stride 16); left/top = 8 U then 8 V; mode in LUMA numbering */
DEV void wave_pred_chroma(uint8_t *dst, const uint8_t *left, const uint8_t *top, int avail, int mode)
{
    const int hl = avail & AV_L, ht = avail & AV_T;
    WAVE_FOR(l)
    {
        if (l < 32)
        {
            int pl = l >> 4, r = (l >> 1) & 7, c = l & 1;       /* plane, row, 4-sample group = one DC quadrant */
            const uint8_t *lf = left + 8*pl, *tp = top + 8*pl;
            uint32_t v;
            if (mode == 0) v = lds32(tp + 4*c);
            else if (mode == 1) v = (uint32_t)lf[r]*0x01010101u;
            else
            {
                int q = (r >> 2)*2 + c, dc;
                if (q == 0) dc = dc_pred(lf, hl, tp, ht, 2);
                else if (q == 1) dc = ht ? dc_pred(lf, 0, tp + 4, 1, 2) : dc_pred(lf, hl, tp, 0, 2);
                else if (q == 2) dc = hl ? dc_pred(lf + 4, 1, tp, 0, 2) : dc_pred(lf, 0, tp, ht, 2);
                else dc = dc_pred(lf + 4, hl, tp + 4, ht, 2);
                v = (uint32_t)dc*0x01010101u;
            }
            lds32_store(dst + 8*pl + 16*r + 4*c, v);
        }
    }
    wave_sync();
}

/*
 * H:1810-1962 h264e_intra_choose_4x4: all nine modes evaluated at once, lane = (mode slot, row).
 * Every predicted sample of every mode (H.264 8.3.1.2; same samples as H:1834-1960) is one of the 13 border samples E[k] (left column
 * bottom-up, corner, top row incl. top-right), a 3-tap value F3[k] = (E[k-1] + 2 E[k] + E[k+1] + 2) >> 2 (border replicated at both
 * ends: that IS the two (a + 3b + 2) >> 2 corner cases), a 2-tap value F2[k] = (E[k] + E[k+1] + 1) >> 1, or the DC.  So the block's 14
 * border lanes build that pool once (their neighbours' samples come through DPP row shifts) and every (mode, row) lane only GATHERS its
 * four samples by the offsets of tables.h k_i4_sel, takes its SAD, the four row SADs of a mode are added inside the quad, and the
 * minimum over the modes -- cost << 4 | slot, so that the earliest slot wins ties exactly like the reference's strict "<" in its test
 * order DC,V,DDL,VL,H,HU,DDR,HD,VR -- is two DPP row rotations and three lane reads.
 * blk = the block's top-left sample in the LDS working picture (row stride as given to i4_lanes_make; row -1 / column -1 hold the neighbours);
 * in = input block (stride 16), pred = LDS output (stride 16); returns mode | cost << 4.
 */
struct I4Scratch { alignas(4) uint8_t pool[48]; };      /* E at 0..13, F3 at 16..28, F2 at 32..43, (DC at 47 in k_i4_sel: taken from a register) */

DEV int i4_slot_mode(int k) { return k == 0 ? 2 : k == 1 ? 0 : k == 2 ? 3 : k == 3 ? 7 : k == 4 ? 1 : k == 5 ? 8 : k == 6 ? 4 : k == 7 ? 6 : 5; }
/* What never changes from block to block, one value per lane, fetched once per macroblock:
 *   sel   lane 4k + y: the four pool offsets of mode slot k, row y (k_i4_sel);
 *   info  lane 4k + y: mode number | neighbours the mode needs << 4 (H:1834-1960: DC none, V DDL VL the top, H HU the left, the rest all
 *         three) | (k < 9) << 8;
 *   eoff  lane k < 14: where border sample E[k] lies relative to the block's top-left sample in the working picture (low half), and where
 *         it comes from when there is no top-right block (high half: its four samples repeat top[3]); E[13] = E[12] */
struct I4Lanes { V64 sel, info, eoff; };
DEV I4Lanes i4_lanes_make(int bstride)
{
    I4Lanes t;
    t.sel = v64_make([&](int l) -> int { const int k = l >> 2; return k < 9 ? (int)k_i4_sel[i4_slot_mode(k)][l & 3] : 0; });
    t.info = v64_make([&](int l) -> int {
        const int k = l >> 2, need = k == 0 ? 0 : k <= 3 ? AV_T : k <= 5 ? AV_L : (AV_T | AV_L | AV_TL);
        return k < 9 ? (i4_slot_mode(k) | (need << 4) | 0x100) : 0;
    });
    t.eoff = v64_make([&](int l) -> int {
        const int k = l > 12 ? 12 : l;
        const int o = k < 4 ? (3 - k)*bstride - 1 : k == 4 ? -bstride - 1 : -bstride + (k - 5);
        const int o2 = k > 8 ? -bstride + 3 : o;
        return (int)(((uint32_t)o & 0xffffu) | ((uint32_t)o2 << 16));
    });
    return t;
}
DEV int wave_i4_choose(const uint8_t *in, uint8_t *pred, int avail, const uint8_t *blk, int mpred, int penalty, I4Scratch &S, const I4Lanes &T)
{
    /* border samples E[0..13] in lanes 0..13 (the lanes behind them read E[12] again: no branch) */
    const int tr = (avail & AV_TR) != 0;
    const V64 e = v64_map(T.eoff, [&](int, int o) -> int { return (int)blk[tr ? (int)(int16_t)(o & 0xffff) : (o >> 16)]; });
    const V64 em = v64_row_shr1(e), ep = v64_row_shl1(e);        /* E[k-1] (E[0] itself for k = 0), E[k+1] */
    v64_each3(e, em, ep, [&](int l, int c, int a, int b) {
        if (l < 13) { S.pool[l] = (uint8_t)c; S.pool[16 + l] = (uint8_t)((a + 2*c + b + 2) >> 2); S.pool[32 + l] = (uint8_t)((c + b + 1) >> 1); }
    });
    int dc;
    {
        /* H:1625-1651: lane k of s4 = E[k] + .. + E[k+3]: the left column's sum in lane 0, the top row's in lane 5 */
        const V64 s2 = v64_map(ep, [&](int l, int b) -> int { return v64_own(e, l) + b; });
        const V64 s4 = v64_map(v64_row_shl1(v64_row_shl1(s2)), [&](int l, int b) -> int { return v64_own(s2, l) + b; });
        const int sl = v64_read(s4, 0), st = v64_read(s4, 5);
        const int hl = (avail & AV_L) != 0, ht = (avail & AV_T) != 0;
        dc = (hl && ht) ? (sl + st + 4) >> 3 : hl ? (sl + 2) >> 2 : ht ? (st + 2) >> 2 : 128;
    }
    wave_sync();
    /* every lane gathers (lanes that stand for no mode gather pool[0]: harmless, and no branch); the DC slot takes the mean */
    const V64 row = v64_map(T.sel, [&](int l, int o) -> int {
        const uint32_t u = (uint32_t)o;
        const uint32_t g = (uint32_t)S.pool[u & 255] | ((uint32_t)S.pool[(u >> 8) & 255] << 8) | ((uint32_t)S.pool[(u >> 16) & 255] << 16) | ((uint32_t)S.pool[u >> 24] << 24);
        return (int)((l >> 2) == 0 ? (uint32_t)dc*0x01010101u : g);
    });
    const V64 qs = v64_quad_sum(v64_map(row, [&](int l, int r) -> int { return (int)sad4_u8(lds32(in + 16*(l & 3)), (uint32_t)r, 0); }));
    /* cost << 4 | slot; a mode whose neighbours are missing cannot win */
    const V64 key = v64_map(qs, [&](int l, int sad) -> int {
        const int f = v64_own(T.info, l), need = (f >> 4) & 15;
        const int ok = (f & 0x100) && (avail & need) == need;
        return ok ? (((sad + ((f & 15) != mpred ? penalty : 0)) << 4) | (l >> 2)) : 0x7fffffff;
    });
    const V64 rm = v64_row_quadmin(key);
    const int bk = imin(imin(v64_read(rm, 0), v64_read(rm, 16)), v64_read(rm, 32));
    const int best = bk & 15, best_cost = bk >> 4;
    v64_each(row, [&](int l, int r) { if ((l >> 2) == best) lds32_store(pred + 16*(l & 3), (uint32_t)r); });
    wave_sync();
    return i4_slot_mode(best) + (best_cost << 4);
}

/* ------------------------------------------------------------------ 4x4 transforms in registers (V16 tiles, wave.h) */

/* one pass of the forward core transform (H:2385-2409) along the quads: d0..d3 at quad positions 0..3 -> outputs k = 0..3 */
DEV V16 v16_fwd_quad(const V16 &d)
{
    const V16 p = v16_quadperm<3, 2, 1, 0>(d);
    const V16 u = v16_map2(d, p, [](int i, int a, int b) -> int { return (i & 2) ? b - a : a + b; });        /* t0 = d0+d3, t2 = d1+d2, t3 = d1-d2, t1 = d0-d3 */
    const V16 A = v16_quadperm<0, 3, 0, 3>(u), Bq = v16_quadperm<1, 2, 1, 2>(u);                               /* t0 t1 t0 t1 | t2 t3 t2 t3 */
    return v16_map2(A, Bq, [](int i, int a, int b) -> int { const int x = i & 3; return x == 0 ? a + b : x == 1 ? 2*a + b : x == 2 ? a - b : a - 2*b; });
}
/* residual in lane 4*y + x -> coefficient in lane 4*k_h + k_v, the reference's storage order (H:2374-2409: rows, then columns; no
 * rounding anywhere and every intermediate fits int16, so the pass order does not matter) */
DEV V16 v16_fwd4x4(const V16 &d) { return v16_fwd_quad(v16_xpose(v16_fwd_quad(d))); }

/* one pass of the inverse transform (H:2436-2489) along the quads, before the int16 truncation of its results */
DEV V16 v16_inv_quad(const V16 &c)
{
    const V16 p = v16_quadperm<2, 3, 0, 1>(c);
    const V16 u = v16_map2(c, p, [](int i, int d, int q) -> int { const int x = i & 3; return x == 0 ? d + q : x == 1 ? (d >> 1) - q : x == 2 ? q - d : q + (d >> 1); });   /* e0 e2 e1 e3 */
    const V16 A = v16_quadperm<0, 2, 2, 0>(u), Bq = v16_quadperm<3, 1, 1, 3>(u);                               /* e0 e1 e1 e0 | e3 e2 e2 e3 */
    return v16_map2(A, Bq, [](int i, int a, int b) -> int { return (i & 2) ? a - b : a + b; });
}
/* dequantized coefficient in lane 4*k_h + k_v -> the residual sample (x, y) in lane 4*x + y (TRANSPOSED: callers address with
 * x = i >> 2, y = i & 3), in the reference's pass order (over k_h first, int16 between the passes, H:2436-2489), rounded: (v + 32) >> 6
 * truncated to int16 like H:2478 */
DEV V16 v16_inv4x4(const V16 &c)
{
    const V16 f = v16_map(v16_inv_quad(v16_xpose(c)), [](int, int v) -> int { return (int16_t)v; });          /* lane 4*k_v + x */
    const V16 g = v16_inv_quad(v16_xpose(f));                                                                   /* lane 4*x + y */
    return v16_map(g, [](int, int v) -> int { return (int16_t)((v + 32) >> 6); });
}

/* ------------------------------------------------------------------ transform / quant */

/*
 * One intra 4x4 block behind its mode decision (H:4790-4811): residual of the input block `bin` (stride 16) against the
 * prediction `pred` (stride 16), forward transform, quantisation (H:2536-2597, no dead zone for intra blocks), inverse transform and
 * reconstruction into `rec` (stride rs) -- all in the registers of one 16-lane tile (the wave's four tiles run the same block; tile 0
 * stores).  Leaves levels and dequantized coefficients in q (mb_write codes the levels when the macroblock ends up intra 4x4).
 * Returns 1 when a level is non-zero.  K: the lane context and the lane's quantiser constants, made once per macroblock (i4q_make).
 */
struct I4Q { V16C C; V16 qm, dqm; int rnd; };
DEV I4Q i4q_make(const uint16_t *qdat)
{
    I4Q k;
    k.C = v16c_make();
    k.qm = v16_make(k.C, [&](int i, int) -> int { return (int)qdat[((i & 1) + ((i >> 2) & 1))*2]; });          /* H:2366 g_idx2quant */
    k.dqm = v16_make(k.C, [&](int i, int) -> int { return (int)qdat[((i & 1) + ((i >> 2) & 1))*2 + 1]; });
    k.rnd = qdat[QD_RND];
    return k;
}
DEV unsigned i4_block_code(const I4Q &K, const uint8_t *bin, const uint8_t *pred, uint8_t *rec, int rs, qblk_t *q)
{
    const V16 d = v16_make(K.C, [&](int i, int) -> int { return (int)bin[16*(i >> 2) + (i & 3)] - (int)pred[16*(i >> 2) + (i & 3)]; });
    const V16 c = v16_fwd4x4_c(K.C, d);
    const int rnd = K.rnd;
    const V16 lev = v16_map2(K.C, c, K.qm, [&](int, int v, int m) -> int { return (mul24(v, m) + (v < 0 ? 0xFFFF - rnd : rnd)) >> 16; });
    const V16 deq = v16_map2(K.C, lev, K.dqm, [](int, int v, int m) -> int { return (int16_t)mul24(v, m); });
    v16_each(K.C, lev, [&](int i, int tile, int v) { if (tile == 0) q->qv[i] = (int16_t)v; });
    v16_each(K.C, deq, [&](int i, int tile, int v) { if (tile == 0) q->dq[i] = (int16_t)v; });
    const unsigned coded = v16_nonzero_mask(v16_map(K.C, lev, [](int, int v) -> int { return (int16_t)v; })) != 0;
    const V16 r = v16_inv4x4_c(K.C, deq);
    v16_each(K.C, r, [&](int i, int tile, int v) { const int x = i >> 2, y = i & 3; if (tile == 0) rec[rs*y + x] = (uint8_t)clip255(v + (int)pred[16*y + x]); });
    wave_sync();
    return coded;
}

/* H:2269-2301 hadamar4_2d (result transposed, every store truncated to int16); uniform, in place */
DEV void hadamard4(int16_t *x)
{
    int16_t tmp[16];
    for (int j = 0; j < 4; j++)
    {
        int a = x[j], b = x[4 + j], c = x[8 + j], d = x[12 + j];
        tmp[4*j + 0] = (int16_t)(a + b + c + d); tmp[4*j + 1] = (int16_t)(a + b - c - d);
        tmp[4*j + 2] = (int16_t)(a - b - c + d); tmp[4*j + 3] = (int16_t)(a - b + c - d);
    }
    for (int i = 0; i < 4; i++)
    {
        int a = tmp[i], b = tmp[4 + i], c = tmp[8 + i], d = tmp[12 + i];
        x[i]     = (int16_t)(a + b + c + d); x[4 + i]  = (int16_t)(a + b - c - d);
        x[8 + i] = (int16_t)(a - b - c + d); x[12 + i] = (int16_t)(a - b + c - d);
    }
}

/* H:2344-2353 h264e_quant_luma_dc (uniform; dc and lev live in LDS) */
DEV void quant_luma_dc(qblk_t *q, int16_t *dc, int16_t *lev, const uint16_t *qdat)
{
    int16_t v[16];
    for (int i = 0; i < 16; i++) v[i] = dc[i];
    hadamard4(v);
    const int quant = (int16_t)qdat[0];
    for (int i = 0; i < 16; i++)
    {
        int r = v[i] < 0 ? (1 << 18) - 0x20000 : 0x20000;
        v[i] = (int16_t)((v[i]*quant + r) >> 18);
        lev[i] = v[i];
    }
    hadamard4(v);
    const int deq = (int16_t)(qdat[1] >> 2);
    for (int i = 0; i < 16; i++) q[i].dq[0] = (int16_t)(v[i]*deq);
    wave_sync();
}

/* H:2355-2364 h264e_quant_chroma_dc */
DEV int quant_chroma_dc(qblk_t *q, int16_t *dc, int16_t *lev, const uint16_t *qdat)
{
    int v[4];
    for (int i = 0; i < 4; i++) v[i] = dc[i];
    const int quant = (int16_t)(qdat[0] << 1), deq = (int16_t)(qdat[1] >> 1);
    for (int k = 0; k < 2; k++)
    {
        int a = v[0], b = v[1], c = v[2], d = v[3];
        v[0] = (int16_t)(a + b + c + d); v[1] = (int16_t)(a - b + c - d);
        v[2] = (int16_t)(a + b - c - d); v[3] = (int16_t)(a - b - c + d);
        if (!k)
            for (int i = 0; i < 4; i++)
            {
                int r = v[i] < 0 ? (1 << 18) - 0xAAAA : 0xAAAA;
                v[i] = (int16_t)((v[i]*quant + r) >> 18);
                lev[i] = (int16_t)v[i];
            }
    }
    for (int i = 0; i < 4; i++) q[i].dq[0] = (int16_t)(v[i]*deq);
    wave_sync();
    return uni((v[0] | v[1] | v[2] | v[3]) != 0);
}

/*
 * mb_write's transform path in ONE register pass (h264-lab.h:2619-2636 + 2638-2681 as H:4423-4488 calls them): residual -> forward
 * transform -> dead-zone tests -> quantiser -> dequantiser -> [DC path] -> inverse transform -> reconstruction, the coefficients never
 * leaving the registers of their 16-lane tile between the steps (wave_xform_quant + wave_recon are four passes with LDS round trips and
 * four wave syncs between them; the reference's order of operations is irrelevant: the stages are pure functions of the block).
 * Blocks are dealt to the passes by 8x8 GROUP -- pass g = the four blocks of group g, one per tile -- because the inter dead zone works
 * on groups (H:2512-2534: a group whose blocks all stay inside the second threshold set is zeroed as a whole): the group test is one
 * ballot of the pass, the per-block test one per tile.  MODE as wave_xform_quant: QMODE_INTER / QMODE_I16 (4 x 4 blocks) or
 * QMODE_CHROMA (2 x 2 blocks = one group).  Leaves levels and dequantised coefficients in q as the reference does (CAVLC reads the
 * levels; the dequantised ones only matter to the stage fixtures, which compare the whole array with the reference's).
 * Returns the non-zero block mask, first block in the highest of the nb bits; *dc_nonzero: quant_chroma_dc's flag (QMODE_CHROMA).
 */
DEV int xq_block(int n, int g, int k) { return n == 4 ? 4*(2*(g >> 1) + (k >> 1)) + 2*(g & 1) + (k & 1) : k; }

template <int MODE> DEV unsigned wave_xform_quant_recon(const uint8_t *inp, const uint8_t *pred, uint8_t *out, int os, qblk_t *q, int16_t *dc, int16_t *lev_dc,
                                                        const uint16_t *qdat, int *dc_nonzero)
{
    constexpr int n = MODE >> 1, i0 = MODE & 1, nb = n*n, NG = n == 4 ? 4 : 1, SUB = 4/V16_TILES;
    constexpr bool dead_zone = MODE == QMODE_INTER || MODE == QMODE_CHROMA;
    const V16C C = v16c_make();              /* the lane's position and transform constants, once for the whole pass (wave.h) */
    /* what a lane needs of the quantiser tables never changes from pass to pass: coefficient position i = lane & 15 */
    const V16 qm = v16_make(C, [&](int i, int) -> int { return (int)qdat[((i & 1) + ((i >> 2) & 1))*2]; });            /* H:2366 g_idx2quant */
    const V16 dqm = v16_make(C, [&](int i, int) -> int { return (int)qdat[((i & 1) + ((i >> 2) & 1))*2 + 1]; });
    const V16 th1 = v16_make(C, [&](int i, int) -> int { return dead_zone ? (int)qdat[QD_THR1 + (i & 7)] : 0; });
    const V16 th2 = v16_make(C, [&](int i, int) -> int { return MODE == QMODE_INTER ? (int)qdat[QD_THR2 + (i & 7)] : 0; });
    const int rnd = qdat[QD_RND];
    V16 keep[NG*SUB];                        /* dequantised coefficients of the modes whose DC takes a path of its own before the reconstruction */
    unsigned mask = 0;
    const auto recon = [&](int g, int s, const V16 &deq) {
        const V16 r = v16_inv4x4_c(C, deq);       /* sample (x, y) of the block in lane 4*x + y */
        v16_each(C, r, [&](int i, int tile, int v) {
            const int b = xq_block(n, g, s*V16_TILES + tile), bx = b & (n - 1), by = b >> (n >> 1), x = i >> 2, y = i & 3;
            out[(size_t)(4*by + y)*os + 4*bx + x] = (uint8_t)clip255(v + (int)pred[64*by + 4*bx + 16*y + x]);
        });
    };
#pragma unroll
    for (int g = 0; g < NG; g++)
    {
        V16 c[SUB], own[SUB];
        int any1 = 0, any2 = 0;
#pragma unroll
        for (int s = 0; s < SUB; s++)
        {
            const V16 d = v16_make(C, [&](int i, int tile) -> int {
                const int b = xq_block(n, g, s*V16_TILES + tile), bx = b & (n - 1), by = b >> (n >> 1), o = 64*by + 4*bx + 16*(i >> 2) + (i & 3);
                return (int)inp[o] - (int)pred[o];
            });
            c[s] = v16_fwd4x4_c(C, d);
            if (i0) v16_each(C, c[s], [&](int i, int tile, int v) { if (i == 0) dc[xq_block(n, g, s*V16_TILES + tile)] = (int16_t)v; });
            if (dead_zone)
            {
                /* H:2491-2534: a coefficient outside [-thr, thr] keeps its block (first set) / its 8x8 group (second set) alive */
                const V16 f1 = v16_map2(C, c[s], th1, [](int i, int v, int t) -> int { return i >= i0 && (unsigned)(v + t) > 2u*(unsigned)t; });
                own[s] = v16_tile_any(C, f1);
                any1 |= v16_any(f1);
                if (MODE == QMODE_INTER) any2 |= v16_any(v16_map2(C, c[s], th2, [](int, int v, int t) -> int { return (unsigned)(v + t) > 2u*(unsigned)t; }));
            }
        }
        const int group_zero = MODE == QMODE_INTER && any1 && !any2;
#pragma unroll
        for (int s = 0; s < SUB; s++)
        {
            const V16 keepb = dead_zone ? (group_zero ? v16_splat(0) : own[s]) : v16_splat(1);       /* per lane: is my block quantised at all? */
            const V16 lev = v16_map3(C, c[s], qm, keepb, [&](int i, int v, int m, int k) -> int {
                return (k && i >= i0) ? (mul24(v, m) + (v < 0 ? 0xFFFF - rnd : rnd)) >> 16 : 0;      /* H:2570-2574; |v| < 2^15, m < 2^16 */
            });
            const V16 deq = v16_map2(C, lev, dqm, [](int, int v, int m) -> int { return (int16_t)mul24(v, m); });
            /* q as the reference leaves it: levels (zero in a zeroed block); dequantised coefficients -- the transform coefficient itself
             * where the quantiser did not run (zeroed blocks, and the DC position the DC path is about to replace) */
            v16_each(C, lev, [&](int i, int tile, int v) { q[xq_block(n, g, s*V16_TILES + tile)].qv[i] = (int16_t)v; });
            const V16 dq_store = v16_map3(C, deq, c[s], keepb, [](int i, int dv, int cv, int k) -> int { return (k && i >= i0) ? dv : cv; });
            v16_each(C, dq_store, [&](int i, int tile, int v) { q[xq_block(n, g, s*V16_TILES + tile)].dq[i] = (int16_t)v; });
            const unsigned nzt = v16_tiles_nonzero(v16_map(C, lev, [](int, int v) -> int { return (int16_t)v; }));
            for (int t = 0; t < V16_TILES; t++) if ((nzt >> t) & 1) mask |= 1u << (nb - 1 - xq_block(n, g, s*V16_TILES + t));
            if (MODE == QMODE_INTER) recon(g, s, deq);
            else keep[g*SUB + s] = deq;
        }
    }
    if (MODE != QMODE_INTER)
    {
        wave_sync();                        /* dc[] is complete */
        if (MODE == QMODE_I16) quant_luma_dc(q, dc, lev_dc, qdat);
        else *dc_nonzero = quant_chroma_dc(q, dc, lev_dc, qdat);
        /* the DC path has left every block's dequantised DC in q[b].dq[0] (both end with a wave sync) */
#pragma unroll
        for (int g = 0; g < NG; g++)
#pragma unroll
            for (int s = 0; s < SUB; s++)
            {
                const V16 dcv = v16_make(C, [&](int i, int tile) -> int { return i == 0 ? (int)q[xq_block(n, g, s*V16_TILES + tile)].dq[0] : 0; });
                recon(g, s, v16_map2(C, keep[g*SUB + s], dcv, [](int i, int a, int d) -> int { return i == 0 ? d : a; }));
            }
    }
    wave_sync();
    return mask;
}

/* ------------------------------------------------------------------ CAVLC (uniform) */

/*
 * H:2775-2949 h264e_vlc_encode.  base[first .. first+maxn-1] (LDS, base 4-byte aligned) scanned in decreasing
 * index order; nctx = left + top nnz context (NNZ_NA = 64 per unavailable side, 17+17 = chroma DC table).
 * Works from the 16-bit mask of non-zero positions, so no per-lane arrays are needed.  Returns TotalCoeff.
 */
/* the CAVLC code tables of tables.h in LDS, one 16-bit entry = length | code << 8 (one ds_read per lookup) */
struct CavlcTab
{
    uint16_t coeff_token[5*17*4];
    uint16_t total_zeros[15*16];
    uint16_t total_zeros_cdc[3*4];
    uint16_t run_before[7*15];
    uint16_t pad;
};
/* one lane per entry; call from every lane of the wave */
DEV void cavlc_tab_load(CavlcTab &t)
{
    WAVE_FOR(l)
    {
        for (int k = l; k < 5*17*4; k += 64) { const uint8_t *e = &k_coeff_token[0][0][0][0] + 2*k; t.coeff_token[k] = (uint16_t)(e[0] | (e[1] << 8)); }
        for (int k = l; k < 15*16; k += 64) { const uint8_t *e = &k_total_zeros[0][0][0] + 2*k; t.total_zeros[k] = (uint16_t)(e[0] | (e[1] << 8)); }
        for (int k = l; k < 3*4; k += 64) { const uint8_t *e = &k_total_zeros_cdc[0][0][0] + 2*k; t.total_zeros_cdc[k] = (uint16_t)(e[0] | (e[1] << 8)); }
        for (int k = l; k < 7*15; k += 64) { const uint8_t *e = &k_run_before[0][0][0] + 2*k; t.run_before[k] = (uint16_t)(e[0] | (e[1] << 8)); }
    }
}
DEV void bw_put_tab(BitW &b, uint16_t e) { bw_put(b, (int)(e & 255), (uint32_t)(e >> 8)); }

DEV int cavlc_block(BitW &b, const CavlcTab &ct, const int16_t *base, int first, int maxn, int nctx)
{
    first = uni(first); maxn = uni(maxn); nctx = uni(nctx);
    /* one LDS read for the whole block: lane p holds coefficient p (wave.h V64); the scalar code below picks coefficients lane by lane
     * instead of one LDS round trip each */
    const V64 cv = v64_make([&](int l) -> int { return l < maxn ? (int)base[first + l] : 0; });
    const uint32_t mask = (uint32_t)v64_nonzero_ballot(cv);
#define COEF(p) v64_read(cv, (p))
    const int total = popc32(mask);
    /* trailing ones: up to three leading (highest position) coefficients of magnitude 1 */
    int t1 = 0;
    uint32_t t1sign = 0;
    {
        uint32_t m = mask;
        while (m && t1 < 3)
        {
            const int p = 31 - clz32(m);
            const int c = COEF(p);
            if (c != 1 && c != -1) break;
            t1sign = (t1sign << 1) | (uint32_t)(c < 0);
            t1++;
            m &= ~(1u << p);
        }
    }
    if (nctx <= 34) nctx = (nctx + 1) >> 1;
    nctx &= 31;
    const int tab = nctx < 2 ? 0 : nctx < 4 ? 1 : nctx < 8 ? 2 : nctx < 17 ? 3 : 4;
    bw_put_tab(b, (uint16_t)uni(ct.coeff_token[(tab*17 + total)*4 + t1]));
    if (!total) return 0;
    if (t1) bw_put(b, t1, t1sign);
    uint32_t m = mask;
    for (int i = 0; i < t1; i++) m &= ~(1u << (31 - clz32(m)));
    int sl = (total > 10 && t1 < 3) ? 1 : 0, firstlev = 1;
    while (m)
    {
        const int p = 31 - clz32(m);
        m &= ~(1u << p);
        const int lv = COEF(p), a = iabs(lv);
        int code = 2*a - 2 + (lv < 0), prefix, nsuf, suf;
        if (firstlev && t1 < 3) code -= 2;
        firstlev = 0;
        if (sl == 0)
        {
            if (code < 14)      { prefix = code; nsuf = 0; suf = 0; }
            else if (code < 30) { prefix = 14; nsuf = 4; suf = code - 14; }
            else                { prefix = 15; nsuf = 12; suf = code - 30; }
        } else
        {
            prefix = code >> sl;
            if (prefix < 15) { nsuf = sl; suf = code - (prefix << sl); }
            else             { prefix = 15; nsuf = 12; suf = code - (15 << sl); }
        }
        bw_put(b, prefix + 1 + nsuf, (1u << nsuf) | (uint32_t)suf);
        if (sl == 0) sl = 1;
        if (a > (3 << (sl - 1)) && sl < 6) sl++;
    }
    if (total < maxn)
    {
        const int top = 31 - clz32(mask);
        int zeros = top + 1 - total;
        if (maxn == 4) bw_put_tab(b, (uint16_t)uni(ct.total_zeros_cdc[(total - 1)*4 + zeros]));
        else           bw_put_tab(b, (uint16_t)uni(ct.total_zeros[(total - 1)*16 + zeros]));
        uint32_t r = mask & ~(1u << top);
        int prev = top;
        while (r && zeros > 0)
        {
            const int p = 31 - clz32(r);
            r &= ~(1u << p);
            const int run = prev - p - 1, zl = zeros > 7 ? 7 : zeros;
            bw_put_tab(b, (uint16_t)uni(ct.run_before[(zl - 1)*15 + run]));
            zeros -= run;
            prev = p;
        }
    }
#undef COEF
    return total;
}

/* ------------------------------------------------------------------ deblocking (on LDS tiles) */

/* normal (bS < 4) luma edge sample, H:1251-1300 / H:1396-1447; p points at q0, s = step across the edge */
DEV void df_luma_normal(uint8_t *p, int s, int alpha, int beta, int tc0)
{
    int p2 = p[-3*s], p1 = p[-2*s], p0 = p[-s], q0 = p[0], q1 = p[s], q2 = p[2*s];
    if (iabs(p0 - q0) < alpha && iabs(p1 - p0) < beta && iabs(q1 - q0) < beta)
    {
        int ap = iabs(p2 - p0) < beta, aq = iabs(q2 - q0) < beta, tc = tc0 + ap + aq;
        int delta = clip3(-tc, tc, (((q0 - p0)*4) + (p1 - q1) + 4) >> 3);
        if (ap) p[-2*s] = (uint8_t)(p1 + clip3(-tc0, tc0, ((p2 + ((p0 + q0 + 1) >> 1)) >> 1) - p1));
        if (aq) p[s]    = (uint8_t)(q1 + clip3(-tc0, tc0, ((q2 + ((p0 + q0 + 1) >> 1)) >> 1) - q1));
        p[-s] = (uint8_t)clip255(p0 + delta);
        p[0]  = (uint8_t)clip255(q0 - delta);
    }
}

/* strong (bS = 4) luma edge sample, H:1302-1394 */
DEV void df_luma_strong(uint8_t *p, int s, int alpha, int beta)
{
    int p3 = p[-4*s], p2 = p[-3*s], p1 = p[-2*s], p0 = p[-s], q0 = p[0], q1 = p[s], q2 = p[2*s], q3 = p[3*s];
    if (iabs(p0 - q0) < alpha && iabs(p1 - p0) < beta && iabs(q1 - q0) < beta)
    {
        int small = iabs(p0 - q0) < ((alpha >> 2) + 2);
        if (small && iabs(p2 - p0) < beta)
        {
            p[-s]   = (uint8_t)((p2 + 2*p1 + 2*p0 + 2*q0 + q1 + 4) >> 3);
            p[-2*s] = (uint8_t)((p2 + p1 + p0 + q0 + 2) >> 2);
            p[-3*s] = (uint8_t)((2*p3 + 3*p2 + p1 + p0 + q0 + 4) >> 3);
        } else
            p[-s] = (uint8_t)((2*p1 + p0 + q1 + 2) >> 2);
        if (small && iabs(q2 - q0) < beta)
        {
            p[0]   = (uint8_t)((q2 + 2*q1 + 2*q0 + 2*p0 + p1 + 4) >> 3);
            p[s]   = (uint8_t)((q2 + q1 + p0 + q0 + 2) >> 2);
            p[2*s] = (uint8_t)((2*q3 + 3*q2 + q1 + q0 + p0 + 4) >> 3);
        } else
            p[0] = (uint8_t)((2*q1 + q0 + p1 + 2) >> 2);
    }
}

/* H:1217-1249 deblock_chroma */
DEV void df_chroma(uint8_t *p, int s, int alpha, int beta, int tc0, int bs)
{
    int p1 = p[-2*s], p0 = p[-s], q0 = p[0], q1 = p[s];
    if (!bs || iabs(p0 - q0) >= alpha || iabs(p1 - p0) >= beta || iabs(q1 - q0) >= beta) return;
    if (bs < 4)
    {
        int tc = tc0 + 1, delta = clip3(-tc, tc, (((q0 - p0)*4) + (p1 - q1) + 4) >> 3);
        p[-s] = (uint8_t)clip255(p0 + delta);
        p[0]  = (uint8_t)clip255(q0 - delta);
    } else
    {
        p[-s] = (uint8_t)((2*p1 + p0 + q1 + 2) >> 2);
        p[0]  = (uint8_t)((2*q1 + q0 + p1 + 2) >> 2);
    }
}

#define YT_STRIDE 24    /* luma tile: rows -4..15, columns -4..15 at (row+4)*24 + col+4 */
#define CT_STRIDE 12    /* chroma tile: rows -2..7, columns -2..7 at (row+2)*12 + col+2 */

/*
 * H:5642-5716 mb_deblock + H:1469-1545 on LDS tiles.  bs[4*e + k]: vertical edge e, rows 4k..4k+3;
 * bs[16 + 4*e + k]: horizontal edge e.  Luma: 16 lanes per edge, the 8 edges in the reference's order;
 * chroma: lanes 0..7 U, 8..15 V.  An edge whose FIRST strength is 4 is strong-filtered over all 16
 * samples (H:1517, H:1534).
 */
/* the deblocking tables of tables.h in LDS (global-memory table reads on every edge were a large part of the filter's time) */
struct DfTab { uint8_t alpha[52], beta[52], tc0[52][3], qpc[52]; };
DEV void df_tab_load(DfTab &t)
{
    WAVE_FOR(l)
    {
        if (l < 52) { t.alpha[l] = k_df_alpha[l]; t.beta[l] = k_df_beta[l]; t.qpc[l] = k_qpc[l]; t.tc0[l][0] = k_df_tc0[l][0]; t.tc0[l][1] = k_df_tc0[l][1]; t.tc0[l][2] = k_df_tc0[l][2]; }
    }
}

DEV void wave_deblock(uint8_t *yt, uint8_t *ct0, uint8_t *ct1, const uint8_t *bs, int qp, int qp_left, int qp_top, const DfTab &D)
{
    for (int dir = 0; dir < 2; dir++)
        for (int e = 0; e < 4; e++)
        {
            const uint8_t *s = bs + 16*dir + 4*e;
            if (!(s[0] | s[1] | s[2] | s[3])) continue;
            int q = e ? qp : dir ? (qp_top + qp + 1) >> 1 : (qp_left + qp + 1) >> 1;
            int alpha = D.alpha[q], beta = D.beta[q];
            if (s[0] != 4 && !alpha) continue;
            WAVE_FOR(l)
            {
                if (l < 16)
                {
                    uint8_t *p = dir ? yt + (4 + 4*e)*YT_STRIDE + 4 + l : yt + (4 + l)*YT_STRIDE + 4 + 4*e;
                    int across = dir ? YT_STRIDE : 1, st = s[l >> 2];
                    if (s[0] == 4) df_luma_strong(p, across, alpha, beta);
                    else if (st) df_luma_normal(p, across, alpha, beta, st < 4 ? D.tc0[q][st - 1] : beta);
                }
            }
            wave_sync();
        }
    const int cq = D.qpc[qp], cql = D.qpc[qp_left], cqt = D.qpc[qp_top];
    for (int dir = 0; dir < 2; dir++)
        for (int e = 0; e < 4; e += 2)
        {
            const uint8_t *s = bs + 16*dir + 4*e;
            int q = e ? cq : dir ? (cqt + cq + 1) >> 1 : (cql + cq + 1) >> 1;
            int alpha = D.alpha[q], beta = D.beta[q];
            if (!(s[0] | s[1] | s[2] | s[3]) || !alpha) continue;
            WAVE_FOR(l)
            {
                if (l < 16)
                {
                    uint8_t *t = (l & 8) ? ct1 : ct0;
                    int i = l & 7, st = s[i >> 1];
                    uint8_t *p = dir ? t + (2 + 2*e)*CT_STRIDE + 2 + i : t + (2 + i)*CT_STRIDE + 2 + 2*e;
                    df_chroma(p, dir ? CT_STRIDE : 1, alpha, beta, (st && st < 4) ? D.tc0[q][st - 1] : 0, st);
                }
            }
            wave_sync();
        }
}

#endif
