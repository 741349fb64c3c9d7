/*
 * stage_harness.c -- TEST INFRASTRUCTURE.  Generator of tests/golden/stages.json: inputs and outputs of the REFERENCE's own
 * per-stage functions, obtained by compiling the reference header into this translation unit (a TU
 * that includes h264-lab.h sees every `static` function).  Built and run in the build container only (`make -C oracle stages`):
 * the reference sources do not travel; the JSON it prints -- data: seeded inputs, the reference's outputs -- is committed.
 * Nothing of the reference is copied here: this file only CALLS
 *   h264e_sad_mb_unlaign_8x8            h264-lab.h:2178     16x16 SAD as four 8x8 quadrant sums
 *   h264e_qpel_interpolate_luma         h264-lab.h:2079     the 16 quarter-sample positions
 *   h264e_qpel_interpolate_chroma       h264-lab.h:2133     eighth-sample bilinear
 *   h264e_transform_sub_quant_dequant   h264-lab.h:2619     forward transform, dead-zone quantiser, dequantiser (4 modes)
 *   h264e_quant_luma_dc / _chroma_dc    h264-lab.h:2344/2355
 *   h264e_transform_add                 h264-lab.h:2638     reconstruction
 *   h264e_vlc_encode                    h264-lab.h:2775     one CAVLC residual block
 *   h264e_intra_choose_4x4              h264-lab.h:1810     intra 4x4 mode decision: nine predictors, tie-breaks, prediction
 *   me_search_diamond + me_mv_set_range h264-lab.h:4973/5181 full-sample diamond search (SAD cache, diagonal probe) + the seven sub-sample probes
 *   df_strength + mb_deblock            h264-lab.h:5532/5642 boundary strengths of a macroblock and its in-loop filter (luma + chroma)
 *   rc_set_qp                           h264-lab.h:5839     the quantiser tables of a QP
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#define MINIH264_IMPLEMENTATION
#include "h264-lab.h"

static uint32_t g_seed = 12345;
static uint32_t rnd(void) { g_seed = g_seed*1664525u + 1013904223u; return g_seed >> 8; }

static void hex(const char *name, const void *p, size_t n, int last)
{
    const uint8_t *b = (const uint8_t *)p;
    size_t i;
    printf("   \"%s\": \"", name);
    for (i = 0; i < n; i++) printf("%02x", b[i]);
    printf("\"%s\n", last ? "" : ",");
}

/* smooth + noise picture, so that interpolation and SAD see edges as well as flat areas */
static void fill_pic(uint8_t *p, int w, int h, int amp)
{
    int x, y;
    for (y = 0; y < h; y++)
        for (x = 0; x < w; x++)
        {
            int v = 128 + (int)(60*((x*7 + y*3) % 32)/32) - 30 + (int)(rnd() % (unsigned)(2*amp + 1)) - amp;
            if ((x / 8 + y / 8) & 1) v += 40;
            p[y*w + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
}

int main(void)
{
    static uint8_t pic[64*64];
    ALIGN(16) static uint8_t blk[16*16] ALIGN2(16), dst[16*16] ALIGN2(16);
    int i, k, first = 1;
    H264E_create_param_t cp;
    h264e_enc_t *enc;
    void *scratch;
    int sizeof_persist = 0, sizeof_scratch = 0;

    memset(&cp, 0, sizeof(cp));
    cp.width = 64; cp.height = 64; cp.gop = 1; cp.vbv_size_bytes = 100000; cp.max_long_term_reference_frames = 0;
    if (H264E_sizeof(&cp, &sizeof_persist, &sizeof_scratch)) return 1;
    enc = (h264e_enc_t *)calloc(1, (size_t)sizeof_persist);
    scratch = calloc(1, (size_t)sizeof_scratch);
    if (!enc || !scratch || H264E_init(enc, &cp)) return 1;

    printf("{\n \"generator\": \"oracle/stage_harness.c against the reference header (h264-lab.h), make -C oracle stages\",\n");

    /* ---- SAD quadrants */
    printf(" \"sad\": [\n");
    for (i = 0; i < 8; i++)
    {
        int sad4[4], tot, ox = (int)(rnd() % 40), oy = (int)(rnd() % 40);
        fill_pic(pic, 64, 64, 4 + 6*i);
        for (k = 0; k < 256; k++) blk[k] = (uint8_t)(pic[(oy + (k >> 4) + ((i & 1) ? 1 : 0))*64 + ox + (k & 15) + ((i & 2) ? 2 : 0)] + (int)(rnd() % 7) - 3);
        tot = h264e_sad_mb_unlaign_8x8(pic + oy*64 + ox, 64, blk, sad4);
        printf("  {\n");
        hex("pic", pic, sizeof(pic), 0);
        hex("blk", blk, 256, 0);
        printf("   \"ox\": %d, \"oy\": %d, \"sad4\": [%d, %d, %d, %d], \"sad\": %d\n  }%s\n", ox, oy, sad4[0], sad4[1], sad4[2], sad4[3], tot, i == 7 ? "" : ",");
    }
    printf(" ],\n");

    /* ---- luma interpolation: all 16 quarter-sample positions at 16x16, the full / half-sample ones at 16x8, 8x16, 8x8 */
    printf(" \"qpel_luma\": [\n");
    fill_pic(pic, 64, 64, 12);
    printf("  {\n");
    hex("pic", pic, sizeof(pic), 0);
    printf("   \"cases\": [\n");
    first = 1;
    for (k = 0; k < 4; k++)
    {
        const int w = (k & 2) ? 8 : 16, h = (k & 1) ? 8 : 16;
        int dx, dy;
        for (dy = 0; dy < 4; dy++)
            for (dx = 0; dx < 4; dx++)
            {
                point_t wh, dxdy;
                const int x0 = 20 + 3*k, y0 = 18 + 5*k;
                if (k && ((dx | dy) & 1)) continue;        /* the quarter-sample positions exist for 16x16 only (h264-lab.h:2114) */
                wh.u32 = 0; dxdy.u32 = 0;
                wh.s.x = (int16_t)w; wh.s.y = (int16_t)h; dxdy.s.x = (int16_t)dx; dxdy.s.y = (int16_t)dy;
                memset(dst, 0, sizeof(dst));
                h264e_qpel_interpolate_luma(pic + y0*64 + x0, 64, dst, wh, dxdy);
                printf("%s    {\"x\": %d, \"y\": %d, \"w\": %d, \"h\": %d, \"dx\": %d, \"dy\": %d,\n ", first ? "" : ",\n", x0, y0, w, h, dx, dy);
                hex("dst", dst, 256, 1);
                printf("    }");
                first = 0;
            }
    }
    printf("\n   ]\n  }\n ],\n");

    /* ---- chroma eighth-sample interpolation */
    printf(" \"qpel_chroma\": [\n");
    fill_pic(pic, 64, 64, 10);
    printf("  {\n");
    hex("pic", pic, sizeof(pic), 0);
    printf("   \"cases\": [\n");
    first = 1;
    for (k = 0; k < 16; k++)
    {
        point_t wh, dxdy;
        const int w = (k & 8) ? 4 : 8, h = (k & 4) ? 4 : 8, dx = (int)(rnd() % 8), dy = (k == 0) ? 0 : (int)(rnd() % 8), x0 = 10 + k, y0 = 30 - k;
        wh.u32 = 0; dxdy.u32 = 0;
        wh.s.x = (int16_t)w; wh.s.y = (int16_t)h; dxdy.s.x = (int16_t)(k == 0 ? 0 : dx); dxdy.s.y = (int16_t)dy;
        memset(dst, 0, sizeof(dst));
        h264e_qpel_interpolate_chroma(pic + y0*64 + x0, 64, dst, wh, dxdy);
        printf("%s    {\"x\": %d, \"y\": %d, \"w\": %d, \"h\": %d, \"dx\": %d, \"dy\": %d,\n ", first ? "" : ",\n", x0, y0, w, h, dxdy.s.x, dxdy.s.y);
        hex("dst", dst, 256, 1);
        printf("    }");
        first = 0;
    }
    printf("\n   ]\n  }\n ],\n");

    /* ---- transform + quantise + dequantise + reconstruct, the four modes, several QPs, residuals from tiny to large */
    printf(" \"quant\": [\n");
    first = 1;
    {
        static const int qps[] = { 10, 22, 26, 33, 40, 51 };
        static const int modes[] = { QDQ_MODE_INTER, QDQ_MODE_INTRA_16, QDQ_MODE_INTRA_4, QDQ_MODE_CHROMA };
        unsigned qi, mi, p_slice;
        for (qi = 0; qi < sizeof(qps)/sizeof(qps[0]); qi++)
            for (p_slice = 0; p_slice < 2; p_slice++)
                for (mi = 0; mi < 4; mi++)
                {
                    const int mode = modes[mi], qp = qps[qi], side = mode >> 1, nblk = (mode == QDQ_MODE_INTRA_4) ? 1 : side*side;
                    const int amp = 2 + (int)((qi*7 + mi*3 + p_slice) % 5)*9;
                    ALIGN(16) static uint8_t inp[16*16] ALIGN2(16), pred[16*16] ALIGN2(16), out[16*16] ALIGN2(16);
                    /* the DC coefficients of the INTRA_16 / CHROMA modes are written in FRONT of the block array (h264-lab.h:2626) */
                    static struct { int16_t dc[16]; quant_t q[16]; } Q, Qsnap;
                    int16_t deq_dc[16];
                    int nz, dcflag = 0;
                    if (mode == QDQ_MODE_INTER && !p_slice) continue;          /* inter blocks exist in P slices only */
                    enc->run_param.qp_min = enc->run_param.qp_max = (uint8_t)qp;
                    enc->slice.type = p_slice ? SLICE_TYPE_P : SLICE_TYPE_I;
                    enc->rc.qp = 0;
                    rc_set_qp(enc, qp);
                    for (k = 0; k < 256; k++)
                    {
                        pred[k] = (uint8_t)(100 + (k & 15)*3 + (int)(rnd() % 9));
                        inp[k] = (uint8_t)(pred[k] + (int)(rnd() % (unsigned)(2*amp + 1)) - amp + (((k >> 6) & 1) ? amp/2 : 0));
                    }
                    memset(&Q, 0, sizeof(Q));
                    memset(deq_dc, 0, sizeof(deq_dc));
                    nz = h264e_transform_sub_quant_dequant(inp, pred, 16, mode, Q.q, enc->rc.qdat[mode == QDQ_MODE_CHROMA ? 1 : 0]);
                    if (mode == QDQ_MODE_INTRA_16) h264e_quant_luma_dc(Q.q, deq_dc, enc->rc.qdat[0]);
                    if (mode == QDQ_MODE_CHROMA) dcflag = h264e_quant_chroma_dc(Q.q, deq_dc, enc->rc.qdat[1]);
                    memcpy(&Qsnap, &Q, sizeof(Q));               /* the reconstruction transforms dq in place */
                    /* reconstruction exactly as mb_write / intra_choose_4x4 call it (h264-lab.h:4428-4433, 4468-4488, 4809-4811) */
                    memcpy(out, pred, sizeof(out));
                    if (mode == QDQ_MODE_INTER) h264e_transform_add(out, 16, pred, Q.q, 4, nz << 16);
                    else if (mode == QDQ_MODE_INTRA_16) h264e_transform_add(out, 16, pred, Q.q, 4, 0xFFFF << 16);
                    else if (mode == QDQ_MODE_INTRA_4) { if (nz & 1) h264e_transform_add(out, 16, pred, Q.q, 1, ~0); }
                    else if (dcflag | nz)
                    {
                        int m = nz, b4;
                        if (dcflag)
                        {
                            for (b4 = 0; b4 < 4; b4++) if (~nz & (8 >> b4)) memset(Q.q[b4].dq + 1, 0, (16 - 1)*sizeof(int16_t));
                            m = 15;
                        }
                        h264e_transform_add(out, 16, pred, Q.q, 2, m << 28);
                    }
                    printf("%s  {\"qp\": %d, \"p_slice\": %u, \"mode\": %d, \"nz\": %d, \"dcflag\": %d,\n", first ? "" : ",\n", qp, p_slice, mode, nz, dcflag);
                    hex("inp", inp, 256, 0);
                    hex("pred", pred, 256, 0);
                    hex("qdat", enc->rc.qdat[mode == QDQ_MODE_CHROMA ? 1 : 0], sizeof(enc->rc.qdat[0]), 0);
                    hex("dc", Qsnap.dc, sizeof(Q.dc), 0);
                    hex("deq_dc", deq_dc, sizeof(deq_dc), 0);
                    hex("q", Qsnap.q, sizeof(quant_t)*(size_t)nblk, 0);
                    hex("out", out, 256, 1);
                    printf("  }");
                    first = 0;
                }
    }
    printf("\n ],\n");

    /* ---- one CAVLC residual block: every table (nC ranges, chroma DC), densities from empty to full, large levels */
    printf(" \"cavlc\": [\n");
    first = 1;
    for (i = 0; i < 96; i++)
    {
        static const int maxn[] = { 16, 15, 4 };
        const int mn = maxn[i % 3], dens = 1 + (i / 3) % 8, big = (i % 7) == 0;
        int16_t q[32], q_in[16];
        uint8_t nzc[3];
        uint8_t buf[256];
        bs_t bs;
        unsigned nbits;
        memset(q, 0, sizeof(q));
        for (k = 0; k < 16; k++)
            if ((int)(rnd() % 9) < dens)
            {
                int v = (int)(rnd() % 3) - 1;
                if ((rnd() % 4) == 0) v = (int)(rnd() % 9) - 4;
                if (big && (rnd() % 3) == 0) v = (int)(rnd() % 4001) - 2000;
                q[k] = (int16_t)v;
            }
        if (i == 0) memset(q, 0, sizeof(q));
        if (mn == 4) { nzc[0] = 17; nzc[2] = 17; }                                    /* the chroma-DC table (h264-lab.h:4477) */
        else { nzc[0] = (uint8_t)(rnd() % 17); nzc[2] = (uint8_t)((rnd() % 5) == 0 ? 64 : rnd() % 17); if ((rnd() % 5) == 0) nzc[0] = 64; }
        nzc[1] = 0xee;
        memcpy(q_in, q, sizeof(q_in));                  /* the function packs the levels into the array it is given */
        memset(buf, 0, sizeof(buf));
        h264e_bs_init_bits(&bs, buf);
        {
            const uint8_t l = nzc[0], t = nzc[2];
            h264e_vlc_encode(&bs, q, mn, nzc + 1);
            nbits = h264e_bs_get_pos_bits(&bs);
            h264e_bs_flush(&bs);
            printf("%s  {\"maxn\": %d, \"left\": %d, \"top\": %d, \"nnz\": %d, \"nbits\": %u,\n", first ? "" : ",\n", mn, l, t, nzc[1], nbits);
        }
        hex("coef", q_in, 32, 0);
        hex("bits", buf, (nbits + 7)/8 + 4, 1);
        printf("  }");
        first = 0;
    }
    printf("\n ],\n");

    /* ---- intra 4x4 mode choice: every neighbour availability, every predicted mode, edges from flat to steep */
    printf(" \"intra4\": [\n");
    first = 1;
    for (i = 0; i < 160; i++)
    {
        ALIGN(16) static uint8_t in4[16*4] ALIGN2(16), pr4[16*4] ALIGN2(16);
        ALIGN(4) uint8_t edge_store[16] ALIGN2(4);          /* [0..3] = L3..L0, [4] = UL, [5..12] = U0..U7 (h264-lab.h:1163-1175) */
        uint8_t edge_in[16];
        const int avail = (i < 16) ? i : (int)(rnd() % 16), mpred = (int)(rnd() % 9), penalty = (i % 5 == 0) ? 0 : (int)(rnd() % 40);
        const int slope = (int)(rnd() % 7) - 3, base = 40 + (int)(rnd() % 160), noise = 1 + (int)(rnd() % 12);
        int ret, y, x;
        for (k = 0; k < 13; k++) { int v = base + slope*(k - 4)*3 + (int)(rnd() % (unsigned)(2*noise + 1)) - noise; edge_store[k] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
        for (k = 13; k < 16; k++) edge_store[k] = 0;
        memset(in4, 0, sizeof(in4)); memset(pr4, 0, sizeof(pr4));
        for (y = 0; y < 4; y++)
            for (x = 0; x < 4; x++)
            {
                int v = base + slope*(x - y)*3 + (int)(rnd() % (unsigned)(2*noise + 1)) - noise;
                if (i % 7 == 3) v = edge_store[5 + x];                          /* exactly vertical: ties between modes */
                if (i % 7 == 5) v = edge_store[3 - y];                          /* exactly horizontal */
                in4[16*y + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
            }
        memcpy(edge_in, edge_store, sizeof(edge_in));
        ret = h264e_intra_choose_4x4(in4, pr4, avail, edge_store + 5, mpred, penalty);
        printf("%s  {\"avail\": %d, \"mpred\": %d, \"penalty\": %d, \"mode\": %d, \"cost\": %d,\n", first ? "" : ",\n", avail, mpred, penalty, ret & 15, ret >> 4);
        hex("edge", edge_in, 13, 0);
        hex("in", in4, 64, 0);
        hex("pred", pr4, 64, 1);
        printf("  }");
        first = 0;
    }
    printf("\n ],\n");

    /* ---- deblocking of one macroblock: strengths from df_strength, then mb_deblock on a 32x32 luma / 16x16 chroma neighbourhood
     * (macroblock at (8,8) / (4,4)); macroblock types, coded-block flags, vectors and QPs vary */
    printf(" \"deblock\": [\n");
    first = 1;
    for (i = 0; i < 36; i++)
    {
        static const int types[] = { -1, 0, 1, 2, 3, 5, 6 };
        static uint8_t Yp[32*32], Up[16*16], Vp[16*16], Yin[32*32], Uin[16*16], Vin[16*16];        /* (U0.. are macros of the reference header) */
        deblock_filter_t df, df2;
        uint8_t dfqp[4], dfqp2[4], dfnz[4], dfnz2[4], strength[32];
        int8_t mbt[4], mbt2[4];
        H264E_io_yuv_t io;
        const int mb_type = types[rnd() % 7], qp = 14 + (int)(rnd() % 36), amp = 2 + (int)(rnd() % 24);
        memset(&df, 0, sizeof(df));
        df.df_qp = dfqp + 1; df.mb_type = mbt + 1; df.df_nzflag = dfnz + 1;
        for (k = 0; k < 4; k++)
        {
            int q2 = qp + (int)(rnd() % 7) - 3;
            dfqp[k] = (uint8_t)(q2 < 10 ? 10 : q2 > 51 ? 51 : q2); mbt[k] = (int8_t)types[rnd() % 7]; dfnz[k] = (uint8_t)(rnd() % 16);
        }
        df.nzflag = (i % 4 == 0) ? 0 : (rnd() & 0x1ffffff) & ((rnd() & 1) ? 0x1ffffff : (rnd() & 0x1ffffff));
        for (k = 0; k < 24; k++)
        {
            df.df_mv[k].s.x = (int16_t)(8 + ((rnd() % 5) == 0 ? (int)(rnd() % 9) - 4 : 0));
            df.df_mv[k].s.y = (int16_t)(-4 + ((rnd() % 5) == 0 ? (int)(rnd() % 9) - 4 : 0));
        }
        fill_pic(Yp, 32, 32, amp); fill_pic(Up, 16, 16, amp/2 + 1); fill_pic(Vp, 16, 16, amp/2 + 1);
        /* blocky content: steps at the 4x4 grid, which is what the filter is for */
        for (k = 0; k < 32*32; k++) Yp[k] = (uint8_t)((Yp[k] >> 1) + 40 + (int)(((k & 31) >> 2) + ((k >> 5) >> 2))*((i % 3) + 1));
        memcpy(Yin, Yp, sizeof(Yp)); memcpy(Uin, Up, sizeof(Up)); memcpy(Vin, Vp, sizeof(Vp));
        /* the strengths this state gives (df_strength updates the state: run it on a copy) */
        df2 = df; memcpy(dfqp2, dfqp, 4); memcpy(mbt2, mbt, 4); memcpy(dfnz2, dfnz, 4);
        df2.df_qp = dfqp2 + 1; df2.mb_type = mbt2 + 1; df2.df_nzflag = dfnz2 + 1;
        memset(strength, 0, sizeof(strength));
        df_strength(&df2, mb_type, 1, strength, 0);
        io.yuv[0] = Yp + 8*32 + 8; io.yuv[1] = Up + 4*16 + 4; io.yuv[2] = Vp + 4*16 + 4;
        io.stride[0] = 32; io.stride[1] = 16; io.stride[2] = 16;
        {
            const int qp_left = dfqp[1], qp_top = dfqp[2];
            mb_deblock(&df, mb_type, qp, 1, 1, &io, 0);
            printf("%s  {\"mb_type\": %d, \"qp\": %d, \"qp_left\": %d, \"qp_top\": %d,\n", first ? "" : ",\n", mb_type, qp, qp_left, qp_top);
        }
        hex("bs", strength, 32, 0);
        hex("y_in", Yin, sizeof(Yin), 0); hex("u_in", Uin, sizeof(Uin), 0); hex("v_in", Vin, sizeof(Vin), 0);
        hex("y_out", Yp, sizeof(Yp), 0); hex("u_out", Up, sizeof(Up), 0); hex("v_out", Vp, sizeof(Vp), 1);
        printf("  }");
        first = 0;
    }
    printf("\n ],\n");

    /* ---- motion search of one partition: a 96x96 reference picture, the macroblock at (32,32) displaced by a known motion */
    {
        /* four reference pictures, shared by the cases */
        static uint8_t refs[4][96*96];
        int r4, x, y;
        printf(" \"diamond_refs\": [\n");
        for (r4 = 0; r4 < 4; r4++)
        {
            const int smooth = 3 + r4;
            for (y = 0; y < 96; y++)
                for (x = 0; x < 96; x++)
                {
                    int a = (x*smooth + y*2) % 64, b = (y*smooth - x + 960) % 48, v;
                    a = a < 32 ? a : 63 - a; b = b < 24 ? b : 47 - b;
                    v = 60 + 3*a + 2*b + (int)(rnd() % 5);
                    refs[r4][y*96 + x] = (uint8_t)(v > 255 ? 255 : v);
                }
            printf("  {\n"); hex("pic", refs[r4], sizeof(refs[r4]), 1); printf("  }%s\n", r4 == 3 ? "" : ",");
        }
        printf(" ],\n");
    printf(" \"diamond\": [\n");
    first = 1;
    for (i = 0; i < 64; i++)
    {
        const uint8_t *refp = refs[i & 3];
        ALIGN(16) static uint8_t cur[256] ALIGN2(16), store[8*256] ALIGN2(16);
        static const int parts[9][4] = { {0,0,16,16}, {0,0,16,8}, {0,8,16,8}, {0,0,8,16}, {8,0,8,16}, {0,0,8,8}, {8,0,8,8}, {0,8,8,8}, {8,8,8,8} };
        const int *pt = parts[i < 16 ? 0 : i % 9], px = pt[0], py = pt[1], w = pt[2], h = pt[3];
        const int tx = (int)(rnd() % 13) - 6, ty = (int)(rnd() % 13) - 6, qx = (int)(rnd() % 4), qy = (int)(rnd() % 4);
        const int qp = 18 + (int)(rnd() % 24), noise = (int)(rnd() % 6);
        point_t mv, mv_pred, wh, dd;
        rectangle_t range;
        pix_t *pbest = 0;
        int ret, min_sad, sx, sy;
        /* the input macroblock: the reference displaced by (tx + qx/4, ty + qy/4) samples, plus noise */
        wh.u32 = 0; wh.s.x = 16; wh.s.y = 16; dd.u32 = 0; dd.s.x = (int16_t)qx; dd.s.y = (int16_t)qy;
        h264e_qpel_interpolate_luma(refp + (32 + ty)*96 + 32 + tx, 96, cur, wh, dd);
        for (k = 0; k < 256; k++) { int v = cur[k] + (noise ? (int)(rnd() % (unsigned)(2*noise + 1)) - noise : 0); cur[k] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
        enc->run_param.encode_speed = (i % 16 == 15) ? 9 : 0;                 /* speed 9: no sub-sample search */
        enc->frame.mv_limit.tl = point(8*4, 8*4); enc->frame.mv_limit.br = point((96 - 16 - 8)*4, (96 - 16 - 8)*4);
        enc->frame.mv_qpel_limit.tl = mv_add(enc->frame.mv_limit.tl, point(4*4, 4*4));
        enc->frame.mv_qpel_limit.br = mv_add(enc->frame.mv_limit.br, point(-4*4, -4*4));
        sx = tx + (int)(rnd() % 7) - 3; sy = ty + (int)(rnd() % 7) - 3;        /* start: near the truth, sometimes on it */
        if (i % 5 == 0) { sx = 0; sy = 0; }
        mv = point((32 + sx)*4, (32 + sy)*4);
        me_mv_set_range(&mv, &range, &enc->frame.mv_limit, 32*4 + py*4);
        mv_pred = point((32 + tx)*4 + (int)(rnd() % 17) - 8, (32 + ty)*4 + (int)(rnd() % 17) - 8);
        wh.s.x = (int16_t)w; wh.s.y = (int16_t)h;
        min_sad = h264e_sad_mb_unlaign_wh(refp + ((mv.s.y >> 2) + py)*96 + (mv.s.x >> 2) + px, 96, cur + py*16 + px, wh) + me_mv_cost(mv, mv_pred, qp);
        if (i % 4 == 1) min_sad = 0x7fffff;
        printf("%s  {\"ref\": %d, \"px\": %d, \"py\": %d, \"w\": %d, \"h\": %d, \"qp\": %d, \"speed\": %d, \"mv_in\": [%d, %d], \"mv_pred\": [%d, %d], \"min_sad_in\": %d,\n"
               "   \"range\": [%d, %d, %d, %d], \"limit\": [%d, %d, %d, %d],\n", first ? "" : ",\n", i & 3, px, py, w, h, qp, enc->run_param.encode_speed,
               mv.s.x, mv.s.y, mv_pred.s.x, mv_pred.s.y, min_sad, range.tl.s.x, range.tl.s.y, range.br.s.x, range.br.s.y,
               enc->frame.mv_limit.tl.s.x, enc->frame.mv_limit.tl.s.y, enc->frame.mv_limit.br.s.x, enc->frame.mv_limit.br.s.y);
        memset(store, 0, sizeof(store));
        ret = me_search_diamond(enc, refp + py*96 + px, cur + py*16 + px, 96, &mv, &range, qp, mv_pred, min_sad, wh, store, &pbest,
                                (w == 16 && h == 16) ? 256 : (w == 8 && h == 16) ? 8 : 128);
        printf("   \"cost\": %d, \"mv\": [%d, %d],\n", ret, mv.s.x, mv.s.y);
        {
            uint8_t blk[256];
            memset(blk, 0, sizeof(blk));
            for (y = 0; y < h; y++) memcpy(blk + 16*y, pbest + 16*y, (size_t)w);
            hex("cur", cur, 256, 0);
            hex("pred", blk, 256, 1);
        }
        printf("  }");
        first = 0;
    }
    }
    enc->run_param.encode_speed = 0;
    printf("\n ]\n}\n");
    free(scratch); free(enc);
    return 0;
}
