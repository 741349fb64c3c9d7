/*
 * enc_selftest.h -- device-level code that only tests and the bench input use: the per-stage test hook (one wave-level stage of the
 * macroblock pipeline on caller-supplied operands) and the synth_v1 clip generator.  Wave64 model like the rest (wave.h).
 */
#ifndef H264E_ENC_SELFTEST_H
#define H264E_ENC_SELFTEST_H
#include "enc_row.h"

/* ------------------------------------------------------------------ per-stage test hook (tests/test_stages.py)
 * Runs ONE of the macroblock pipeline's wave-level stages on caller-supplied operands, so that each can be compared with the
 * reference's own function of the same stage (tests/golden/stages.json, made by oracle/stage_harness.c):
 *   1 SAD quadrants (wave_sad_ref_q)           in: picture 64x64 | block 16x16         args: x, y, window      out: int32 sad4[4], sum
 *   2 luma quarter-sample (wave_interp_luma)   in: picture 64x64                       args: x, y, w, h, dx, dy, window   out: 16x16 (stride 16)
 *   3 chroma bilinear (wave_interp_chroma)     in: picture 64x64 (used as U and V)     args: x, y, w, h, dx, dy           out: 16x16: U cols 0-7, V cols 8-15
 *   4 transform/quant/dequant/recon            in: inp 256 | pred 256 | qdat 42 x u16  args: mode              out: int32 nz, dcflag | qblk_t q[16] | i16 dc[16] | i16 lev[16] | recon 256
 *   5 CAVLC block (cavlc_block)                in: int16 coef[16]                      args: first, maxn, nctx out: int32 nnz, nbits | bytes
 *   8 motion search of one partition (diamond)  in: picture 96x96 | macroblock 16x16      args: px, py, w, h, mv x/y, pred x/y, min_sad, qp, speed, range[4], limit[4], window
 *                                                out: int32 cost, mv x, mv y | prediction 16x16 (stride 16)
 *   7 deblock one macroblock (wave_deblock)     in: luma tile 20x24 | U tile 10x12 | V tile 10x12 | bs 32   args: qp, qp_left, qp_top   out: the three tiles
 *   6 intra 4x4 mode choice (wave_i4_choose)   in: edge 13 (L3..L0, UL, U0..U7) | block 4x4 stride 16   args: avail, mpred, penalty   out: int32 mode, cost | prediction 4x4 stride 16
 * `window` = 1 reads the reference samples through the LDS window like the macroblock loop, 0 through the HBM path.
 */
struct StageLds
{
    alignas(16) uint8_t win[WIN_W*WIN_STRIDE + 16];
    alignas(16) uint8_t a[256], b[256], o[256];
    alignas(16) qblk_t q[16];
    alignas(4) int16_t dc[16], lev[16], coef[16];
    alignas(4) uint16_t qdat[42];
    CavlcTab ct;
    I4Scratch i4s;
    DfTab df;
    alignas(16) uint8_t yt[20*YT_STRIDE];
    alignas(16) uint8_t ctile[2][10*CT_STRIDE];
    alignas(4) uint8_t bs[32];
    alignas(4) uint8_t nb[5*24];         /* the block's neighbourhood as intra4_choose keeps it: row stride 24, block at row 1, column 4 */
};
#define STAGE_IN_MAX (96*96 + 256)
#define STAGE_NARGS 24
#define STAGE_OUT_MAX (16 + 16*64 + 32 + 32 + 256)
DEV void stage_selftest(StageLds &S, RowLds &L, int stage, const GLOBAL_AS uint8_t *in, const int *args, GLOBAL_AS uint8_t *out)
{
    int a[STAGE_NARGS];
    for (int i = 0; i < STAGE_NARGS; i++) a[i] = uni(args[i]);
    Plane P = { (const gu8 *)in, 64, 64, 64 };
    RefView R;
    R.P = P; R.win = (const lu8 *)S.win; R.has_win = 0; R.wx0 = 0; R.wy0 = 0; R.dep = 0; R.nmbx = 4; R.nmby = 4; R.vw = WIN_W; R.vh = WIN_W;
    R.far = 0; R.fail = 0; R.slice_row = 0; R.nslices = 0; R.spin_limit = 0;
    GLOBAL_AS int32_t *oi = (GLOBAL_AS int32_t *)out;
    if (stage == 1 || stage == 2)
    {
        const int x = a[0], y = a[1], window = stage == 1 ? a[2] : a[6];
        if (window) { R.has_win = 1; R.wx0 = x - WIN_M; R.wy0 = y - WIN_M; wave_load_window(S.win, P, R.wx0, R.wy0, 0); wave_sync(); }
        if (stage == 1)
        {
            int s4[4];
            WAVE_FOR(l) { lds32_store(S.b + 4*l, gload32((const gu8 *)in + 4096 + 4*l)); }
            wave_sync();
            const int tot = wave_sad_ref_q(R, x, y, S.b, s4);
            if (wave_lane() == 0) { oi[0] = s4[0]; oi[1] = s4[1]; oi[2] = s4[2]; oi[3] = s4[3]; oi[4] = tot; }
        } else
        {
            WAVE_FOR(l) { lds32_store(S.o + 4*l, 0u); }
            wave_sync();
            wave_interp_luma(R, 0, 0, mvmk(4*x + a[4], 4*y + a[5]), a[2], a[3], S.o);
            wave_sync();
            WAVE_FOR(l) { gstore32((gu8 *)out + 4*l, lds32(S.o + 4*l)); }
        }
    } else if (stage == 3)
    {
        WAVE_FOR(l) { lds32_store(S.o + 4*l, 0u); }
        wave_sync();
        wave_interp_chroma(R, P, P, 0, 0, mvmk(8*a[0] + a[4], 8*a[1] + a[5]), a[2], a[3], S.o);
        wave_sync();
        WAVE_FOR(l) { gstore32((gu8 *)out + 4*l, lds32(S.o + 4*l)); }
    } else if (stage == 4)
    {
        const int mode = a[0], side = mode >> 1;
        WAVE_FOR(l)
        {
            lds32_store(S.a + 4*l, gload32((const gu8 *)in + 4*l));
            lds32_store(S.b + 4*l, gload32((const gu8 *)in + 256 + 4*l));
            if (l < 21) lds32_store((uint8_t *)S.qdat + 4*l, gload32((const gu8 *)in + 512 + 4*l));
            if (l < 8) { lds32_store((uint8_t *)S.dc + 4*l, 0u); lds32_store((uint8_t *)S.lev + 4*l, 0u); }
            for (int k = l; k < 256; k += 64) lds32_store((uint8_t *)S.q + 4*k, 0u);
        }
        wave_sync();
        /* the product's paths: mb_write's fused pass (transform -> dead zone -> quantiser -> DC path -> reconstruction) for the inter, intra 16x16
         * and chroma modes, intra4_choose's block coder for the intra 4x4 mode; q (levels + dequantised coefficients), the DC levels and the
         * reconstruction are what the reference's functions leave (h264-lab.h:2619-2681, 4428-4433, 4468-4488, 4809-4811) */
        unsigned nz = 0;
        int dcflag = 0;
        WAVE_FOR(l) { lds32_store(S.o + 4*l, lds32(S.b + 4*l)); }
        wave_sync();
        if (mode == QMODE_INTER) nz = wave_xform_quant_recon<QMODE_INTER>(S.a, S.b, S.o, 16, S.q, S.dc, S.lev, S.qdat, (int *)0);
        else if (mode == QMODE_I16) nz = wave_xform_quant_recon<QMODE_I16>(S.a, S.b, S.o, 16, S.q, S.dc, S.lev, S.qdat, (int *)0);
        else if (mode == QMODE_CHROMA) nz = wave_xform_quant_recon<QMODE_CHROMA>(S.a, S.b, S.o, 16, S.q, S.dc, S.lev, S.qdat, &dcflag);
        else nz = i4_block_code(i4q_make(S.qdat), S.a, S.b, S.o, 16, S.q);
        wave_sync();
        WAVE_FOR(l)
        {
            for (int k = l; k < 256; k += 64) gstore32((gu8 *)out + 8 + 4*k, lds32((const uint8_t *)S.q + 4*k));
            if (l < 8) { gstore32((gu8 *)out + 8 + 1024 + 4*l, lds32((const uint8_t *)S.dc + 4*l)); gstore32((gu8 *)out + 8 + 1056 + 4*l, lds32((const uint8_t *)S.lev + 4*l)); }
        }
        wave_sync();
        WAVE_FOR(l) { gstore32((gu8 *)out + 8 + 1088 + 4*l, lds32(S.o + 4*l)); }
        if (wave_lane() == 0) { oi[0] = (int32_t)nz; oi[1] = dcflag; }
        (void)side;
    } else if (stage == 5)
    {
        cavlc_tab_load(S.ct);
        WAVE_FOR(l) { if (l < 8) lds32_store((uint8_t *)S.coef + 4*l, gload32((const gu8 *)in + 4*l)); }
        wave_sync();
        BitW b;
        b.acc = 0; b.nacc = 0; b.pos = 0; b.cap = 60; b.overflow = 0; b.buf = (GLOBAL_AS uint32_t *)(out + 8);
        const int nnz = cavlc_block(b, S.ct, S.coef, a[0], a[1], a[2]);
        const uint32_t nbits = bw_bits(b);
        if (b.nacc) bw_put(b, 32 - b.nacc, 0);
        if (wave_lane() == 0) { oi[0] = nnz; oi[1] = (int32_t)nbits; }
    } else if (stage == 6)
    {
        WAVE_FOR(l)
        {
            if (l < 13)
            {
                const uint8_t e = in[l];
                if (l < 4) S.nb[24*(4 - l) + 3] = e;            /* L3..L0: the column left of the block, bottom-up */
                else S.nb[3 + (l - 4)] = e;                     /* UL, U0..U7: the row above */
            }
            if (l < 16) lds32_store(S.a + 4*l, gload32((const gu8 *)in + 16 + 4*l));
            if (l < 16) lds32_store(S.o + 4*l, 0u);
        }
        wave_sync();
        const int res = wave_i4_choose(S.a, S.o, a[0], S.nb + 24 + 4, a[1], a[2], S.i4s, i4_lanes_make(24));
        wave_sync();
        WAVE_FOR(l) { if (l < 16) gstore32((gu8 *)out + 8 + 4*l, lds32(S.o + 4*l)); }
        if (wave_lane() == 0) { oi[0] = res & 15; oi[1] = res >> 4; }
    } else if (stage == 7)
    {
        const int ny = 20*YT_STRIDE, nc = 10*CT_STRIDE;
        df_tab_load(S.df);
        WAVE_FOR(l)
        {
            for (int k = l; k < ny; k += 64) S.yt[k] = in[k];
            for (int k = l; k < nc; k += 64) { S.ctile[0][k] = in[ny + k]; S.ctile[1][k] = in[ny + nc + k]; }
            if (l < 32) S.bs[l] = in[ny + 2*nc + l];
        }
        wave_sync();
        wave_deblock(S.yt, S.ctile[0], S.ctile[1], S.bs, a[0], a[1], a[2], S.df);
        wave_sync();
        WAVE_FOR(l)
        {
            for (int k = l; k < ny; k += 64) out[k] = S.yt[k];
            for (int k = l; k < nc; k += 64) { out[ny + k] = S.ctile[0][k]; out[ny + nc + k] = S.ctile[1][k]; }
        }
    } else if (stage == 8)
    {
        /* the macroblock at (32,32) of a 96x96 reference picture, as row_step sets a macroblock up for inter_choose */
        h264e_geom_t Gs;
        MbCtx m;
        Plane P8 = { (const gu8 *)in, 96, 96, 96 };
        Gs.width = Gs.W = 96; Gs.height = Gs.H = 96; Gs.nmbx = Gs.nmby = 6; Gs.nmb = 36; Gs.cropping = 0;
        Gs.lim_x0 = a[15]; Gs.lim_y0 = a[16]; Gs.lim_x1 = a[17]; Gs.lim_y1 = a[18];
        m.G = &Gs; m.speed = a[10]; m.slice_type = 0; m.x = 2; m.y = 2; m.num = 14; m.qp = a[9];
        m.lambda_mv = k_lambda_mv_q4[a[9]];
        m.rv = R; m.rv.P = P8; m.rv.nmbx = 6; m.rv.nmby = 6;
        if (a[19]) { m.rv.has_win = 1; m.rv.win = (const lu8 *)L.win; m.rv.wx0 = 32 - WIN_M; m.rv.wy0 = 32 - WIN_M; wave_load_window(L.win, P8, m.rv.wx0, m.rv.wy0, 0); }
        WAVE_FOR(l) { lds32_store(L.mb[0].inp + 4*l, gload32((const gu8 *)in + 96*96 + 4*l)); lds32_store(L.gtest[0] + 4*l, 0u); }
        wave_sync();
        const rect_t range = { a[11], a[12], a[13], a[14] };
        /* the search is lane-group code (wave.h): group a[20] runs it, the other three idle */
        GRP_EACH(grp)
        {
            if (grp == (a[20] & 3))
            {
                mv32 mv = mvmk(a[4], a[5]);
                const int cost = diamond_g(L, L.mb[0], m, a[0], a[1], mv, range, mvmk(a[6], a[7]), a[8], a[2], a[3], L.gtest[0] + 16*a[1] + a[0], L.gscr);
                L.gcost[0] = cost; L.gcost[1] = mvx(mv); L.gcost[2] = mvy(mv);
            }
        }
        wave_sync();
        WAVE_FOR(l) { gstore32((gu8 *)out + 16 + 4*l, lds32(L.gtest[0] + 4*l)); }
        if (wave_lane() == 0) { oi[0] = L.gcost[0]; oi[1] = L.gcost[1]; oi[2] = L.gcost[2]; }
    }
}

/* synth_v1 generator, one sample per call */
DEV uint32_t sv_h32(uint32_t a)
{
    a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
    return a;
}
DEV int sv_lattice(int32_t ix, int32_t iy, uint32_t seed) { return (int)(sv_h32((uint32_t)ix*0x9E3779B1u ^ (uint32_t)iy*0x85EBCA77u ^ seed) & 255); }
DEV int sv_tex(int32_t X, int32_t Y, uint32_t seed, int lg)
{
    int32_t c = 1 << lg, ix = X >> lg, iy = Y >> lg, fx = X & (c - 1), fy = Y & (c - 1);
    int32_t a = sv_lattice(ix, iy, seed), b = sv_lattice(ix + 1, iy, seed), cc = sv_lattice(ix, iy + 1, seed), d = sv_lattice(ix + 1, iy + 1, seed);
    int32_t top = a*(c - fx) + b*fx, bot = cc*(c - fx) + d*fx;
    return (top*(c - fy) + bot*fy + (1 << (2*lg - 1))) >> (2*lg);
}
DEV uint8_t sv_sample(int w, int h, int t, uint32_t seed, int idx)
{
    const int32_t OFF = 1 << 20;
    if (idx < w*h)
    {
        int x = idx % w, y = idx / w, fw = w/8 > 32 ? w/8 : 32, fh = h/6 > 32 ? h/6 : 32;
        int fx0 = (w/2 + ((10*t) >> 2)) % (w - fw), fy0 = h/3, v;
        if (x >= fx0 && x < fx0 + fw && y >= fy0 && y < fy0 + fh) v = sv_tex(4*x - 10*t + OFF, 4*y + OFF, seed + 1, 5);
        else v = (sv_tex(4*x + 5*t + OFF, 4*y + 3*t + OFF, seed, 6)*3 >> 2) + 32;
        v += (int)(sv_h32((uint32_t)x ^ ((uint32_t)y << 12) ^ ((uint32_t)t << 24) ^ (uint32_t)(seed*7919u)) % 5) - 2;
        return (uint8_t)clip255(v);
    }
    idx -= w*h;
    const int cw = w/2, ch = h/2, pl = idx >= cw*ch;
    if (pl) idx -= cw*ch;
    int x = idx % cw, y = idx / cw;
    if (!pl) return (uint8_t)(128 + ((sv_tex(8*x + 5*t + OFF, 8*y + 3*t + OFF, seed + 2, 7) - 128) >> 2));
    return (uint8_t)(128 - ((sv_tex(8*x + 5*t + OFF, 8*y + 3*t + OFF, seed + 3, 7) - 128) >> 3));
}

#endif
