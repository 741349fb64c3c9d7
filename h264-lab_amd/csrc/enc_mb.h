/*
 * enc_mb.h -- the macroblock pipeline in the wave64 model (device code, gfx950): one wavefront encodes one
 * macroblock row, macroblock after macroblock (row_step), behind the wavefront dependency of the row above.
 *
 * Reproduces, decision for decision, mb_encode and everything below it in /root/reference/src/h264-lab.h
 * ("H:n"): inter_choose_mode (H:5283), me_search_diamond (H:4973), intra_choose_16x16/4x4 (H:4876, H:4723),
 * mb_write (H:4378), df_strength/mb_deblock (H:5535, H:5642).  Layout is ours: neighbour state of the row
 * above comes from per-macroblock records in HBM, the left neighbour lives in LDS, bits go to a per-row
 * buffer that the frame's finalizer workgroup splices (skip runs are resolved there).
 */
#ifndef H264E_ENC_MB_H
#define H264E_ENC_MB_H

#include <stddef.h>
#include "enc_kernels.h"

/* predictor context of the motion search (H:3646-3671 saves and restores it around every partitioning; here every partition type works on
 * its own copy): the left column, the top-left column and the top row (+ top-right) of 4x4-block vectors */
struct GCtx { mv32 mv_left[4], mv_tl[4], mv_top[8]; };
/* sub-pel scratch of the four partition types: 4 planes of w*h bytes each (16x16, 16x8, 8x16, 8x8) */
#define GSCR_OFF(t) ((t) == 0 ? 0 : (t) == 1 ? 1024 : (t) == 2 ? 1536 : 2048)
#define GSCR_BYTES 2304

/*
 * What one macroblock hands from the SEARCH side of the pipeline (neighbour records + input + motion search) to the RECONSTRUCTION
 * side (intra decisions, transform / CAVLC, deblocking, stores).  Two of them per row (macroblock x uses mb[x & 1]): with two
 * wavefronts per row (h264e_kernels.hip) the search wave fills the buffer of macroblock x + 1 while the reconstruction wave still
 * works from the buffer of x.
 */
struct MbBuf
{
    alignas(16) uint8_t inp[256];
    alignas(16) uint8_t inp_c[128];
    alignas(16) uint8_t pred[256];                  /* luma prediction of the decision so far (inter, then 16x16 intra when that wins) */
    alignas(4) uint8_t pix_top[36];                 /* 16 Y, 8 U, 8 V of the macroblock above + 4 Y of the one above-right */
    alignas(4) uint8_t ptop[96];                    /* pending bottom lines of the macroblock above (h264e_mbpend_t), fetched with its record */
    mv32 mv_top[8];
    uint8_t nnz_top[8];
    int8_t i4_top[4];
    uint32_t df_nz_top;
    int top_type, top_qp;
    mv32 mv[16], mvd[16];
    /* result of the inter decision (H:5283-5524): type -1 skip, 0..3 partition type; an I slice leaves type 0, cost 0x7fffffff */
    int type, cost, used_cand;
    mv32 mv_skip_pred;
};

struct RowLds
{
    /* ---- search side: predictor context carried along the row (written by the decision of macroblock x, read by the search of x + 1) */
    mv32 mv_left[4], mv_tl[4];
    /* ---- reconstruction side: carried from macroblock to macroblock along the row */
    uint8_t nnz_left[8];
    int8_t i4_left[4];
    alignas(4) uint8_t pix_left[32];
    uint8_t pix_tl[4];
    uint32_t df_nzflag;
    mv32 df_mv[25];
    int left_type, left_qp;
    alignas(4) uint8_t strip_y[16*4];               /* deblocked columns 12..15 of the left macroblock (luma), 6..7 (chroma) */
    alignas(4) uint8_t strip_c[2][8*4];             /* columns 4..7 of the left macroblock's chroma (final except column 7) */
    alignas(4) uint8_t brec[64];                    /* record of this macroblock for the row below, assembled here */
    BitW bw;
    int skip_run, lead_skips, coded_any;
    /* ---- both sides */
    int far_reads[3];                               /* reference accesses of this row that left the valid window (search / reconstruction side / 8x8 search helper) */
    int far_fail[3];                                /* a dynamic wait behind such an access gave up (enc_kernels.h rv_wait_rect): -1 expired, -2 producer aborted */
    int16_t slice_row[H264E_MAX_SLICES + 2];       /* this frame's slice start rows (copy of the task's) */
    unsigned long long prof[4][32], prof_last[4], prof_c0, prof_w0;     /* -DH264E_STAMPS diagnostic build only */
    /* hand-off words of the two-wave pipeline (h264e_kernels.hip): monotonic counters "macroblocks done" per stage, and a stop code */
    int f_noskip, f_bound, f_inter, f_decided, f_wdone, f_stop;
    int early_bound;                                /* an upper bound of the inter cost, known right after the candidate evaluation (f_bound) */
    /* three waves per row (latency-bound launches): the search wave hands the 8x8 partition type to a helper wave */
    int f_t3req, f_t3done;                          /* macroblocks for which the 8x8 search was requested / is done */
    int f_front, d_type[2];                         /* four waves per row: macroblocks whose mb_recon_front is done (the fourth wave takes mb_recon_back), their final types */
    int t3_sad_best, t3_lim[4];                     /* the request: start cost, vector limits (the start vector below) */
    mv32 t3_mv_best;

    MbBuf mb[2];
    alignas(4) uint8_t trec[72];                    /* staged record of the macroblock above (+ 8 bytes of the one above-right) */

    /* ---- reconstruction side, per macroblock */
    int8_t i4_mode[16];
    alignas(4) uint8_t bs[32];
    I4Scratch i4s;
    uint8_t nzctx[12];
    alignas(16) uint8_t pred_c[128];
    alignas(16) uint8_t i16pred[256];               /* 16x16 intra prediction under test */
    alignas(16) uint8_t tt[256];
    alignas(16) uint8_t i4rec[17*24];               /* intra 4x4 working picture: row 0 / column 0 = neighbours */
    alignas(16) uint8_t ytile[20*YT_STRIDE];
    alignas(16) uint8_t ctile[2][10*CT_STRIDE];
    alignas(16) qblk_t qy[16];
    qblk_t qu[4], qv[4];
    int16_t dcy[16], dcu[4], dcv[4], lev_dcy[16], lev_dcu[4], lev_dcv[4];

    /* ---- tables of the frame (read-only after row_begin) */
    CavlcTab cavlc;
    DfTab dftab;
    int qconst[6];                                  /* this frame's decision constants: lambda_mv, lambda_q4, skip_thr, skip_thr_i4, lambda_i4, lambda_i16 */
    uint16_t qdat[2][42];                           /* this frame's quantizer tables, copied from the task */

    /* ---- search side, per macroblock.  LAST in the struct: the intra-only kernel variant never touches these and allocates the
     * struct only up to here (ROWLDS_INTRA_BYTES), which is what lets twice as many of its workgroups fit a CU */
    mv32 part_mv[4][4], part_mvd[4][4];
    GCtx gctx[4];                                   /* the motion search's predictor context, one copy per partition type (lane group) */
    int gcost[4], gnum[4];                          /* cost and number of partitions of every partition type searched */
    alignas(16) uint8_t win[WIN_W*WIN_STRIDE + 16];  /* reference luma window around the current macroblock */
    alignas(16) uint8_t skip_pred[256];
    alignas(16) uint8_t skip_pred_c[128];           /* chroma of the early-skip test (the reconstruction side predicts chroma again for itself) */
    alignas(16) uint8_t gtest[4][256];              /* prediction of every partition type searched */
    alignas(16) uint8_t gscr[GSCR_BYTES];           /* sub-pel search: the full-sample block and the three half-sample planes of the partition a group works on */

};
#define ROWLDS_INTRA_BYTES offsetof(RowLds, part_mv)

struct MbCtx
{
    const h264e_geom_t *G;
    int speed, slice_type;                          /* of the frame (copied from the row's task: values, no pointer into it) */
    mv32 clu[2];                                    /* speculated mv_clusters pair of the frame ... */
    const mv32 *clu_per_mb;                         /* ... or the exact per-macroblock trajectory in HBM (NULL: none) */
    Plane ref[3];
    RefView rv;                                     /* reference luma through the LDS window */
    gu8 *dec[3];
    int x, y, num, avail, type, cost, i16_mode, cropped, used_cand;
    int slice_top;                                  /* first macroblock row of its slice: nothing above is used or filtered */
    mv32 mv_skip_pred;
    unsigned nz_mask;
    int qp;
    int lambda_mv, lambda_q4, skip_thr, skip_thr_i4, lambda_i4, lambda_i16;     /* this QP's decision constants */
};

struct rect_t { int x0, y0, x1, y1; };

DEV int in_rect(mv32 v, const rect_t &r) { return mvy(v) >= r.y0 && mvy(v) <= r.y1 && mvx(v) >= r.x0 && mvx(v) <= r.x1; }
DEV mv32 clip_rect(mv32 v, const rect_t &r) { return mvmk(imin(imax(mvx(v), r.x0), r.x1), imin(imax(mvy(v), r.y0), r.y1)); }
DEV mv32 mb_abs(const MbCtx &m, mv32 v) { return mvadd(v, mvmk(m.x*64, m.y*64)); }
DEV int mv_cost(const MbCtx &m, mv32 v, mv32 pred)                                  /* H:4952 */
{
    return MUL_LAMBDA(se_len(mvx(v) - mvx(pred)) + se_len(mvy(v) - mvy(pred)), m.lambda_mv);
}
DEV rect_t mv_limit(const MbCtx &m) { rect_t r = { m.G->lim_x0, m.G->lim_y0, m.G->lim_x1, m.G->lim_y1 }; return r; }
DEV rect_t mv_qlimit(const MbCtx &m) { rect_t r = { m.G->lim_x0 + 16, m.G->lim_y0 + 16, m.G->lim_x1 - 16, m.G->lim_y1 - 16 }; return r; }

/* ------------------------------------------------------------------ MV prediction (uniform) */

DEV int med3(int a, int b, int c) { return imax(imin(imax(a, b), c), imin(a, b)); }

/* H:3696-3715 me_mv_medianpredictor_put, 4x4-block units */
DEV void mvp_put_arr(mv32 *mv_left, mv32 *mv_tl, mv32 *mv_top, int x, int y, int w, int h, mv32 mv)
{
    mv_tl[y] = mv_top[x + w - 1];
    for (int i = 1; i < h; i++) mv_tl[y + i] = mv;
    for (int i = 0; i < h; i++) mv_left[y + i] = mv;
    for (int i = 0; i < w; i++) mv_top[x + i] = mv;
}
DEV void mvp_put(RowLds &L, MbBuf &B, int x, int y, int w, int h, mv32 mv) { mvp_put_arr(L.mv_left, L.mv_tl, B.mv_top, x, y, w, h, mv); }

/* H:3720-3872 me_mv_medianpredictor_get */
DEV mv32 mvp_get_arr(const mv32 *mv_left, const mv32 *mv_tl, const mv32 *mv_top, int flag, int x, int y, int w, int h)
{
    int type = 1;
    mv32 a = mv_left[y], b = mv_top[x], c = mv_top[x + w], d = mv_tl[y], ret = 0;
    if (!x)
    {
        if (!(flag & AV_L)) a = MV_NA;
        if (!(flag & AV_TL)) d = MV_NA;
    }
    if (!y)
    {
        if (!(flag & AV_T))
        {
            b = MV_NA;
            if (x + w < 4) c = MV_NA;
            if (x > 0) d = MV_NA;
        }
        if (!(flag & AV_TL) && !x) d = MV_NA;
        if (!(flag & AV_TR) && x + w == 4) c = MV_NA;
    }
    if (x + w == 4 && (!(flag & AV_TR) || y)) c = d;
#define OK(v) ((v) != MV_NA)
    if (OK(a) && !OK(b) && !OK(c)) type = 2;
    else if (!OK(a) && OK(b) && !OK(c)) type = 3;
    else if (!OK(a) && !OK(b) && OK(c)) type = 4;
    if (w == 2 && h == 4)
    {
        if (x == 0) { if (OK(a)) type = 2; } else { if (OK(c)) type = 4; }
    } else if (w == 4 && h == 2)
    {
        if (y == 0) { if (OK(b)) type = 3; } else { if (OK(a)) type = 2; }
    }
    if (type == 2) { if (OK(a)) ret = a; }
    else if (type == 3) { if (OK(b)) ret = b; }
    else if (type == 4) { if (OK(c)) ret = c; }
    else if (!(OK(b) || OK(c))) { if (OK(a)) ret = a; }
    else
    {
        if (!OK(a)) a = 0;
        if (!OK(b)) b = 0;
        if (!OK(c)) c = 0;
        ret = mvmk(med3(mvx(a), mvx(b), mvx(c)), med3(mvy(a), mvy(b), mvy(c)));
    }
#undef OK
    return ret;
}
DEV mv32 mvp_get(const RowLds &L, const MbBuf &B, const MbCtx &m, int x, int y, int w, int h) { return (mv32)uni(mvp_get_arr(L.mv_left, L.mv_tl, B.mv_top, m.avail, x, y, w, h)); }

/* ------------------------------------------------------------------ motion search */

/* H:5181-5193 me_mv_set_range */
DEV void set_range(mv32 &pnt, rect_t &range, const rect_t &limit, int mby_q)
{
    rect_t r = limit;
    r.y0 = (int16_t)imax(r.y0, mby_q - 63*4);
    r.y1 = (int16_t)imin(r.y1, mby_q + 63*4);
    pnt = clip_rect(pnt, r);
    mv32 tl = clip_rect(mvadd(pnt, mvmk(-32*4, -32*4)), r), br = clip_rect(mvadd(pnt, mvmk(32*4, 32*4)), r);
    range.x0 = mvx(tl); range.y0 = mvy(tl); range.x1 = mvx(br); range.y1 = mvy(br);
}

/*
 * H:4973-5176 me_search_diamond for the w x h partition at (px,py) of the macroblock; mv is absolute for
 * the macroblock (quarter-pel), so the block sits at (px,py) + (mv >> 2) in the reference picture.
 * The uint16 SAD cache with its 0xffff sentinel is observable behaviour.
 * dst (LDS, stride 16) receives the prediction of the returned vector.
 * Runs inside a LANE-GROUP section (wave.h): the four partition types of a macroblock are searched side by side, one 16-lane group
 * each, so every "scalar" of this function is a per-lane value that agrees inside the group; the block's dwords are dealt to the
 * group's lanes 16 per pass.  scr = the group's sub-pel scratch (4 planes of w*h bytes).
 */
DEV int diamond_g(RowLds &L, const MbBuf &B, const MbCtx &m, int px, int py, mv32 &mv, const rect_t &range, mv32 mv_pred, int min_sad, int w, int h, uint8_t *dst, uint8_t *scr)
{
    const RefView &R = m.rv;
    const uint8_t *b = B.inp + 16*py + px;
    /* the reference's uint16 cache[8] (H:4993-4997): the SADs of the centre's four neighbours (c0..c3: +x, -x, +y, -y) and of the previous
     * centre's (p0..p3), 0xffff = not evaluated -- the truncation to 16 bits and the sentinel are observable */
    int c0, c1, c2, c3, p0, p1, p2, p3;
#define DX(d) ((d) == 0 ? 4 : (d) == 1 ? -4 : 0)
#define DY(d) ((d) == 2 ? 4 : (d) == 3 ? -4 : 0)
#define SEL4(d, a0, a1, a2, a3) ((d) == 0 ? (a0) : (d) == 1 ? (a1) : (d) == 2 ? (a2) : (a3))
    int cost;
    mv32 v;
    const int g = w >> 2, npass = (g*h) >> 4;
    for (;;)
    {
        /* H:4999-5051, one CENTRE per iteration instead of one direction: the reference scans the four neighbours of the centre in the order
         * d0, d0+1, d0+2, d0+3 (d0 = the direction it arrived from; 0 at the start), skips those outside the range or already in the cache,
         * caches every cost it takes and moves to the FIRST neighbour that improves on the centre.  With the four SADs taken in one batch that
         * is a priority pick: the improving neighbour with the smallest scan position wins, the neighbours scanned up to it enter the cache. */
        int d0 = 0, dir_prev = -1;
        c0 = c1 = c2 = c3 = p0 = p1 = p2 = p3 = 0xffff;
        for (;;)
        {
            /* (the centre lies inside the range -- set_range clips the start vector into it, a move only goes to a neighbour inside it -- so a
             * neighbour can leave it on its own side only: one comparison each instead of in_rect's four) */
            const int e0 = mvx(mv) + 4 <= range.x1 && c0 == 0xffff, e1 = mvx(mv) - 4 >= range.x0 && c1 == 0xffff;
            const int e2 = mvy(mv) + 4 <= range.y1 && c2 == 0xffff, e3 = mvy(mv) - 4 >= range.y0 && c3 == 0xffff;
            if (!(e0 | e1 | e2 | e3)) break;
            const int want = e0 | (e1 << 1) | (e2 << 2) | (e3 << 3);
            int s4[4];
            const int cx = px + (mvx(mv) >> 2), cy = py + (mvy(mv) >> 2);
            if (rv_inside(R, cx - 1, cy - 1, cx + w, cy + h))
            {
                const lu8 *base = rv_ptr(R, cx, cy);
                grp_sum4([&](int i, int *sv) {
                    uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
                    for (int k = 0; k < npass; k++)
                    {
                        const int d = i + 16*k, r = d >> (g >> 1), c4 = d & (g - 1);
                        const uint32_t in4 = lds32(b + 16*r + 4*c4);
                        const lu8 *p = base + r*WIN_STRIDE + 4*c4;
                        /* all four neighbours lie inside the window: issue the loads together (one LDS wait instead of four);
                         * sums of directions that are not wanted are simply not looked at */
                        const uint32_t a0 = lds32u(p + 1), a1 = lds32u(p - 1), a2 = lds32u(p + WIN_STRIDE), a3 = lds32u(p - WIN_STRIDE);
                        t0 = sad4_u8(a0, in4, t0); t1 = sad4_u8(a1, in4, t1); t2 = sad4_u8(a2, in4, t2); t3 = sad4_u8(a3, in4, t3);
                    }
                    sv[0] = (int)t0; sv[1] = (int)t1; sv[2] = (int)t2; sv[3] = (int)t3;
                }, s4);
            } else
            {
                rv_wait_rect_g(R, cy - 1, cx + w, cy + h);
                grp_sum4([&](int i, int *sv) {
                    uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
                    for (int k = 0; k < npass; k++)
                    {
                        const int d = i + 16*k, r = d >> (g >> 1), c4 = d & (g - 1);
                        const uint32_t in4 = lds32(b + 16*r + 4*c4);
                        if (want & 1) t0 = sad4_u8(ref_load4(R.P, cx + 4*c4 + 1, cy + r), in4, t0);
                        if (want & 2) t1 = sad4_u8(ref_load4(R.P, cx + 4*c4 - 1, cy + r), in4, t1);
                        if (want & 4) t2 = sad4_u8(ref_load4(R.P, cx + 4*c4, cy + r + 1), in4, t2);
                        if (want & 8) t3 = sad4_u8(ref_load4(R.P, cx + 4*c4, cy + r - 1), in4, t3);
                    }
                    sv[0] = (int)t0; sv[1] = (int)t1; sv[2] = (int)t2; sv[3] = (int)t3;
                }, s4);
            }
            /* SAD + vector cost of the four neighbours */
            int c4v[4];
            grp_eval4([&](int d) -> int { return mv_cost(m, mvadd(mv, mvmk(DX(d), DY(d))), mv_pred); }, c4v);
            const int b0 = s4[0] + c4v[0], b1 = s4[1] + c4v[1], b2 = s4[2] + c4v[2], b3 = s4[3] + c4v[3];
            /* scan position of every direction, 4 = "does not improve"; the first improving one wins */
            const int o0 = (0 - d0) & 3, o1 = (1 - d0) & 3, o2 = (2 - d0) & 3, o3 = (3 - d0) & 3;
            const int wk = imin(imin((e0 && b0 < min_sad) ? o0 : 4, (e1 && b1 < min_sad) ? o1 : 4), imin((e2 && b2 < min_sad) ? o2 : 4, (e3 && b3 < min_sad) ? o3 : 4));
            const int lim = wk < 4 ? wk : 3;
            if (e0 && o0 <= lim) c0 = b0 & 0xffff;
            if (e1 && o1 <= lim) c1 = b1 & 0xffff;
            if (e2 && o2 <= lim) c2 = b2 & 0xffff;
            if (e3 && o3 <= lim) c3 = b3 & 0xffff;
            if (wk == 4) break;
            {
                /* H:5021-5040: move to the winner; the cache follows (the previous centre becomes the neighbour behind, and the
                 * neighbour the last two moves enclose -- the "corner" -- is known from the centre before) */
                const int win = (d0 + wk) & 3;
                const int corner = dir_prev >= 0 ? SEL4(win, p0, p1, p2, p3) : 0xffff;
                const int back = win ^ 1, side = dir_prev >= 0 ? (dir_prev ^ 1) : -1;
                p0 = c0; p1 = c1; p2 = c2; p3 = c3;
                c0 = back == 0 ? (min_sad & 0xffff) : side == 0 ? corner : 0xffff;
                c1 = back == 1 ? (min_sad & 0xffff) : side == 1 ? corner : 0xffff;
                c2 = back == 2 ? (min_sad & 0xffff) : side == 2 ? corner : 0xffff;
                c3 = back == 3 ? (min_sad & 0xffff) : side == 3 ? corner : 0xffff;
                min_sad = SEL4(win, b0, b1, b2, b3);
                mv = mvadd(mv, mvmk(DX(win), DY(win)));
                dir_prev = win;
                d0 = win;
            }
        }

        const int pri = c3 >= c2 ? 2 : 3, sec = c1 >= c0 ? 0 : 1;
        v = mvadd(mv, mvmk(DX(pri) + DX(sec), DY(pri) + DY(sec)));
        if (in_rect(v, range))
        {
            cost = grp_sad_ref(R, px + (mvx(v) >> 2), py + (mvy(v) >> 2), b, w, h) + mv_cost(m, v, mv_pred);
            if (cost < min_sad)
            {
                mv = v;
                min_sad = cost;
                continue;       /* H:5074 goto restart */
            }
        }
        break;
    }
    STAMP(L, 25);
#undef SEL4
#undef DX
#undef DY

#ifdef H264E_ABLATE
    if (H264E_ABLATE == 3) { grp_interp_luma(R, px, py, mv, w, h, dst); return min_sad; }
#endif
    if (!(m.speed < 9 && in_rect(mv, mv_qlimit(m))))
    {
        grp_interp_luma(R, px, py, mv, w, h, dst);
        return min_sad;
    }
    {
        /* H:5083-5174: seven sub-pel probes around the full-pel winner -- half-pels towards the cheaper vertical and the
         * cheaper horizontal neighbour (02, 20), their diagonal (22), and the quarter-pels between them (01, 10, 11, 12).
         * The reference always evaluates all seven, in this order, keeping the first strict minimum; here every lane builds
         * its 4 samples of all seven (plus the full-pel block) in one pass and the seven SADs are reduced together.  The
         * full-sample block and the three half-sample planes go to the group's scratch; the quarter-sample candidates are
         * their rounded averages and are formed again only for the winner. */
        mv32 pq = mvmk(0, -1), sq = mvmk(-1, 0);
        uint32_t ms1 = c1, ms2 = c3;
        if (c3 >= c2) { pq = mvmk(0, 1); ms2 = c2; }
        if (c1 >= c0) { sq = mvmk(1, 0); ms1 = c0; }
        if (ms2 > ms1) { mv32 sw = sq; sq = pq; pq = sw; }
        const mv32 vdg = mvadd(pq, sq);
        const mv32 v02 = mvadd(mv, mvadd(pq, pq)), v01 = mvadd(mv, pq), v20 = mvadd(mv, mvadd(sq, sq)), v10 = mvadd(mv, sq);
        const mv32 v11 = mvadd(mv, vdg), v22 = mvadd(mv, mvadd(vdg, vdg)), v12 = mvadd(mv, mvadd(pq, vdg));
        int s8[8];
        const int fx0 = px + (mvx(mv) >> 2), fy0 = py + (mvy(mv) >> 2), plane = w*h;
        const bool inside = rv_inside(R, fx0 - 5, fy0 - 3, fx0 + w + 4, fy0 + h + 3);      /* every probe is within one sample of mv */
        if (!inside) rv_wait_rect_g(R, fy0 - 3, fx0 + w + 4, fy0 + h + 3);
        /* the 2x2 integer cell that holds the three half-sample positions: vdg = (+-1, +-1) says on which side of (x,y) it lies */
        const int hp_ox = mvx(vdg) < 0 ? 1 : 0, hp_oy = mvy(vdg) < 0 ? 1 : 0, hp_pq_vertical = mvx(pq) == 0;
#define AVG4(x, y) (((x) | (y)) - ((((x) ^ (y)) >> 1) & 0x7f7f7f7fu))                    /* per-byte (x + y + 1) >> 1 */
        grp_sum8([&](int i, int *sv) {
            uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0;
            /* a lane owns `npass` CONSECUTIVE rows of one 4-sample column group (not every 16th dword as elsewhere): consecutive rows share
             * five of the six filtered window rows the half-sample planes need (enc_kernels.h halfpel3_rows) */
            const int c4 = i & (g - 1), r0 = (i >> (g >> 1))*npass;
            const auto probe = [&](int r, uint32_t q00, uint32_t q02, uint32_t q20, uint32_t q22) {
                const int o = w*r + 4*c4;
                const uint32_t in4 = lds32(b + 16*r + 4*c4);
                const uint32_t q01 = AVG4(q00, q02), q10 = AVG4(q00, q20), q11 = AVG4(q02, q20), q12 = AVG4(q22, q02);
                lds32_store(scr + o, q00); lds32_store(scr + plane + o, q02); lds32_store(scr + 2*plane + o, q20); lds32_store(scr + 3*plane + o, q22);
                t0 = sad4_u8(q02, in4, t0); t1 = sad4_u8(q01, in4, t1); t2 = sad4_u8(q20, in4, t2);
                t3 = sad4_u8(q10, in4, t3); t4 = sad4_u8(q11, in4, t4); t5 = sad4_u8(q22, in4, t5);
                t6 = sad4_u8(q12, in4, t6);
            };
            if (inside)
            {
                /* all three half-sample planes from one pass over the window */
                halfpel3_rows(rv_ptr(R, fx0 + 4*c4 - hp_ox, fy0 + r0 - hp_oy), hp_ox, hp_oy, npass, [&](int k, const hp4_t &hp) {
                    probe(r0 + k, hp.x, hp_pq_vertical ? hp.z : hp.y, hp_pq_vertical ? hp.y : hp.z, hp.w);
                });
            } else
            {
                for (int k = 0; k < npass; k++)
                {
                    const int r = r0 + k;
#define IP(vv) interp_luma4(R, false, px + (mvx(vv) >> 2) + 4*c4, py + (mvy(vv) >> 2) + r, mvx(vv) & 3, mvy(vv) & 3)
                    const uint32_t q00 = IP(mv), q02 = IP(v02), q20 = IP(v20), q22 = IP(v22);
#undef IP
                    probe(r, q00, q02, q20, q22);
                }
            }
            sv[0] = (int)t0; sv[1] = (int)t1; sv[2] = (int)t2; sv[3] = (int)t3; sv[4] = (int)t4; sv[5] = (int)t5; sv[6] = (int)t6;
        }, s8);
        wave_sync();
        int best = -1, c8[8];
        mv32 vbest = mv;
        /* the seven vector costs (H:4952-4956: lambda * (se bits of dx + se bits of dy) >> 4): the probes lie at 0, 1 and 2 steps of pq and of sq
         * from mv, one of the two vertical and the other horizontal, so their cost is made of three code lengths per axis */
        {
            const int pax = hp_pq_vertical, p_step = pax ? mvy(pq) : mvx(pq), s_step = pax ? mvx(sq) : mvy(sq);
            const int p_mv = pax ? mvy(mv) : mvx(mv), s_mv = pax ? mvx(mv) : mvy(mv), p_pr = pax ? mvy(mv_pred) : mvx(mv_pred), s_pr = pax ? mvx(mv_pred) : mvy(mv_pred);
            int P[4], S[4];
            grp_eval4([&](int k) -> int { return se_len((int16_t)(p_mv + k*p_step) - p_pr); }, P);
            grp_eval4([&](int k) -> int { return se_len((int16_t)(s_mv + k*s_step) - s_pr); }, S);
            c8[0] = MUL_LAMBDA(P[2] + S[0], m.lambda_mv); c8[1] = MUL_LAMBDA(P[1] + S[0], m.lambda_mv); c8[2] = MUL_LAMBDA(P[0] + S[2], m.lambda_mv);
            c8[3] = MUL_LAMBDA(P[0] + S[1], m.lambda_mv); c8[4] = MUL_LAMBDA(P[1] + S[1], m.lambda_mv); c8[5] = MUL_LAMBDA(P[2] + S[2], m.lambda_mv);
            c8[6] = MUL_LAMBDA(P[2] + S[1], m.lambda_mv);
        }
#define TRY(j, vv) { const int cst = s8[j] + c8[j]; if (cst < min_sad) { min_sad = cst; vbest = (vv); best = j; } }
        TRY(0, v02) TRY(1, v01) TRY(2, v20) TRY(3, v10) TRY(4, v11) TRY(5, v22) TRY(6, v12)
#undef TRY
        /* the winner's samples: a stored plane, or the rounded average of two of them */
        const int pa = best < 0 ? 0 : best == 0 ? 1 : best == 1 ? 0 : best == 2 ? 2 : best == 3 ? 0 : best == 4 ? 1 : best == 5 ? 3 : 3;
        const int pb = best == 1 ? 1 : best == 3 ? 2 : best == 4 ? 2 : best == 6 ? 1 : pa;
        GRP_FOR(i)
        {
            for (int k = 0; k < npass; k++)
            {
                const int d = i + 16*k, r = d >> (g >> 1), c4 = d & (g - 1), o = w*r + 4*c4;
                const uint32_t x = lds32(scr + pa*plane + o), y = lds32(scr + pb*plane + o);
                lds32_store(dst + 16*r + 4*c4, AVG4(x, y));          /* the average of a plane with itself is the plane */
            }
        }
#undef AVG4
        wave_sync();
        mv = vbest;
    }
    STAMP(L, 26);
    return min_sad;
}

/*
 * One partition type of H:5283-5524's partition loop, run by one lane group: t = 0 16x16, 1 16x8, 2 8x16, 3 8x8.  The partitions of a
 * type are a serial chain through their vector predictors (the group's own copy of the predictor context, L.gctx[t]); the four types
 * are independent of each other: every one starts from the macroblock's predictor context (H:3646-3671).  Leaves the type's cost in
 * L.gcost[t], its vectors in L.part_mv[t] / L.part_mvd[t] and its prediction in L.gtest[t].
 */
DEV void search_type(RowLds &L, const MbBuf &B, const MbCtx &m, int t, mv32 mv_best, int sad_best0, const rect_t &lim, int skip_cost)
{
    GCtx &X = L.gctx[t];
    int imv = 0, part_sad = MUL_LAMBDA(t == 0 ? 1 : t == 3 ? 12 : 4, m.lambda_q4);
    const int w = (t & 2) ? 8 : 16, h = (t & 1) ? 8 : 16;
    int px = 0, py = 0;
    uint8_t *test = L.gtest[t], *scr = L.gscr + GSCR_OFF(t);
    for (;;)
    {
        rect_t range;
        mv32 mvabs = mb_abs(m, mv_best);
        int sad_best = sad_best0;
        const mv32 mvp = mvp_get_arr(X.mv_left, X.mv_tl, X.mv_top, m.avail, px >> 2, py >> 2, w >> 2, h >> 2);
        if (!t) set_range(mvabs, range, lim, m.y*64 + py*4);
        else
        {
            mvabs = mvround(mb_abs(m, mvp));
            set_range(mvabs, range, lim, m.y*64 + py*4);
            sad_best = grp_sad_ref(m.rv, px + (mvx(mvabs) >> 2), py + (mvy(mvabs) >> 2), B.inp + 16*py + px, w, h)
                     + mv_cost(m, mvabs, mb_abs(m, mvp));
        }
        STAMP(L, 24);
        part_sad += diamond_g(L, B, m, px, py, mvabs, range, mb_abs(m, mvp), sad_best, w, h, test + 16*py + px, scr);
        const mv32 mv = mvsub(mvabs, mvmk(m.x*64, m.y*64));
        L.part_mvd[t][imv] = mvsub(mv, mvp);
        L.part_mv[t][imv++] = mv;
        mvp_put_arr(X.mv_left, X.mv_tl, X.mv_top, px >> 2, py >> 2, w >> 2, h >> 2, mv);
        wave_sync();
        STAMP(L, 27);
        /* two waves per row: the 16x16 cost is final after this group's first step -- a tighter bound of the macroblock's inter cost than
         * the one inter_choose announced (the decision is the minimum over the types, or the skip vector's cost when that minimum lies
         * above its SAD), for the intra 4x4 cut-off of the reconstruction wave while the other types are still being searched */
        if (t == 0) L.early_bound = imax(part_sad, skip_cost);
        px = (px + w) & 15;
        if (!px)
        {
            py = (py + h) & 15;
            if (!py) break;
        }
    }
    L.gcost[t] = part_sad;
    L.gnum[t] = imv;
}

/* H:5224-5257 mb_inter_partition */
DEV void partition_hints(const int sad[4], int mode[4])
{
    int sum = sad[0] + sad[1] + sad[2] + sad[3];
    int slope = iabs((sad[0] - sad[2]) + (sad[1] - sad[3])) - iabs((sad[0] - sad[1]) + (sad[2] - sad[3]));
    int skew = iabs(sad[3] - sad[0]) - iabs(sad[2] - sad[1]);
    if (slope > (sum >> 4)) mode[1] = 1;
    if (slope < -(sum >> 4)) mode[2] = 1;
    if (iabs(skew) > (sum >> 4) && iabs(slope) <= (sum >> 4)) mode[3] = 1;
}

/* H:4915-4947 interpolate_chroma: every partition of the current type, both planes -> L.pred_c */
DEV void predict_chroma_inter(const MbBuf &B, const MbCtx &m, uint8_t *pred_c)
{
    int w = (m.type & 2) ? 4 : 8, h = (m.type & 1) ? 4 : 8, part = 0, x = 0, y = 0;
    if (m.type == -1) w = h = 8;
    for (;; part++)
    {
        wave_interp_chroma(m.rv, m.ref[1], m.ref[2], x, y, mb_abs(m, B.mv[part]), w, h, pred_c + 16*y + x);
        x = (x + w) & 7;
        if (!x)
        {
            y = (y + h) & 7;
            if (!y) break;
        }
    }
}

/*
 * Chroma half of the early-skip test, H:5322-5349.  For cropped edge macroblocks the reference copies
 * the padded 8x8 input into mb_pix_store (H:5333), which at that point is the buffer holding the chroma
 * prediction (ptest after the swap of H:5316): rows 0..3 of the prediction are overwritten by the input
 * copy before the SAD is taken -- first by U, then again by V.  Reproduced arithmetically:
 * row r < 4 of plane c is compared with input row 2r + c instead of the prediction.
 */
DEV int skip_chroma_ok(const MbBuf &B, const MbCtx &m, const uint8_t *pred_c)
{
    const int thr = m.skip_thr;
    for (int c = 0; c < 2; c++)
    {
        int sad = wave_sum([&](int l) -> int {
            if (l >= 16) return 0;
            int r = l >> 1, g = l & 1;
            uint32_t a = lds32(B.inp_c + 16*r + 8*c + 4*g);
            uint32_t p = (m.cropped && r < 4) ? lds32(B.inp_c + 16*(2*r + c) + 8*c + 4*g) : lds32(pred_c + 16*r + 8*c + 4*g);
            return (int)sad4_u8(a, p, 0);
        });
        if (sad >= thr) return 0;
    }
    return 1;
}

/* H:5283-5524 inter_choose_mode */
/* sig: what the two-wave pipeline wants to hear before the decision is complete (enc_row.h NoSignals / h264e_kernels.hip SearchSignals):
 * noskip() once the early-skip test has failed, bound(u) as soon as an upper bound u of the final inter cost is known; helper(): a
 * third wave searches the 8x8 partition type (t3_request hands it over, t3_wait returns false when the row is being stopped) */
template <class SIG> DEV void inter_choose(RowLds &L, MbBuf &B, MbCtx &m, SIG sig)
{
    int prefer[4] = { 1, 0, 0, 0 };      /* constant indices only after unrolling: stays in registers */
    const RefView &R = m.rv;
    const int bx = m.x*16, by = m.y*16;
    int sad, sad_skip = 0x7FFFFFFF, sad_best = 0x7FFFFFFF, cand_cost_best = 0, j = 0, ncand = 0, sad4[4];
    mv32 mv_best = MV_NA;
    LaneArr cand;                           /* start candidates (H:5360-5386), register resident */
    cand.clear();

    /* H:3877-3890 skip predictor */
    const mv32 mv_pred16 = mvp_get(L, B, m, 0, 0, 4, 4);
    m.mv_skip_pred = 0;
    if (!(~m.avail & (AV_L | AV_T)) && L.mv_left[0] != 0 && B.mv_top[0] != 0) m.mv_skip_pred = mv_pred16;
    const mv32 mv_skip = m.mv_skip_pred, mv_skip_a = mb_abs(m, mv_skip);

    STAMP(L, 2);
    if (in_rect(mv_skip_a, mv_qlimit(m)))
    {
        wave_interp_luma(R, 0, 0, mv_skip_a, 16, 16, L.skip_pred);
        sad_skip = wave_sad_lds_q(B.inp, L.skip_pred, sad4);
        if (imax(imax(sad4[0], sad4[1]), imax(sad4[2], sad4[3])) < m.skip_thr)
        {
            m.type = -1;
            B.mv[0] = mv_skip;
            m.cost = 0;
            predict_chroma_inter(B, m, L.skip_pred_c);
            if (skip_chroma_ok(B, m, L.skip_pred_c))
            {
                wave_copy_wh(B.pred, L.skip_pred, 16, 16);
                return;
            }
        }
        if (m.speed < 1) partition_hints(sad4, prefer);
        mv_best = mvround(mv_skip);
        cand.set(ncand++, mv_best);
        if (!((mvx(mv_skip) | mvy(mv_skip)) & 3))
        {
            sad_best = sad_skip;
            cand_cost_best = mv_cost(m, mv_skip, mv_pred16);
            j = 1;
        }
    }
    sig.noskip();               /* not an early skip: the intra candidates will be wanted */

    STAMP(L, 3);
    m.used_cand = 1;
    cand.set(ncand++, mv_pred16);
    cand.set(ncand++, 0);                                                    /* H:3895-3914 */
    if ((m.avail & AV_L) && L.mv_left[0] != MV_NA) cand.set(ncand++, L.mv_left[0]);
    if ((m.avail & AV_T) && B.mv_top[0] != MV_NA) cand.set(ncand++, B.mv_top[0]);
    if ((m.avail & AV_TR) && B.mv_top[4] != MV_NA) cand.set(ncand++, B.mv_top[4]);
    if (m.x <= 0) cand.set(ncand++, mvmk(8*4, 0));
    if (m.y <= 0) cand.set(ncand++, mvmk(0, 8*4));
    {
        /* the speculated mv_clusters pair: frame-constant (registers of the row's task copy) or this macroblock's entry of the
         * exact trajectory in HBM.  (Values, not a pointer: the frame-constant pair is not in global memory.) */
        mv32 c0 = m.clu[0], c1 = m.clu[1];
        if (m.clu_per_mb)
        {
            const GLOBAL_AS mv32 *clu = (const GLOBAL_AS mv32 *)m.clu_per_mb + 2*m.num;
            c0 = clu[0]; c1 = clu[1];
        }
        cand.set(ncand++, c0);
        cand.set(ncand++, c1);
    }
    {   /* H:5198-5218 round to full-pel, drop duplicates */
        int k = 1;
        cand.set(0, mvround(cand.get(0)));
        for (int n = 1; n < ncand; n++)
        {
            const mv32 v = mvround(cand.get(n));
            if (!cand.has(v, k)) cand.set(k++, v);      /* one compare + ballot instead of a scan of the list */
        }
        ncand = k;
    }
    wave_sync();

    const rect_t lim = mv_limit(m);
    {
        /* H:5388-5409: the start candidates are compared in list order with a strict "<" on SAD + vector cost, every one that is tried also
         * votes on the partition types to search (H:5224-5257).  Four candidates at a time, one per lane group: the winner is the smallest
         * (SAD + cost) << 4 | list position (the earliest of equals, like the strict "<"), starting from the skip vector's figures when it
         * is a full-sample vector (j = 1); the votes are OR-ed over all groups. */
        int key = 0x7fffffff, h1 = 0, h2 = 0, h3 = 0;
        for (int base = j; base < ncand; base += 4)
        {
            GRP_EACH(gq)
            {
                const int jj = base + gq;
                const mv32 cj = (mv32)cand.get_any(jj), va = mb_abs(m, cj);          /* (read with every lane active: a lane-crossing read) */
                if (jj < ncand)
                {
                    if (in_rect(va, lim))
                    {
                        int s4[4], hint[4] = { 0, 0, 0, 0 };
                        const int sad = grp_sad_ref_q(R, bx + (mvx(cj) >> 2), by + (mvy(cj) >> 2), B.inp, s4);
                        if (m.speed < 1) partition_hints(s4, hint);
                        h1 |= hint[1]; h2 |= hint[2]; h3 |= hint[3];
                        key = imin(key, ((sad + mv_cost(m, cj, mv_pred16)) << 4) | jj);
                    }
                }
            }
        }
        /* gather: the groups' votes and the smallest key of the four groups (a group's lanes agree) */
        prefer[1] |= wave_ballot([&](int) -> int { return h1; }) != 0;
        prefer[2] |= wave_ballot([&](int) -> int { return h2; }) != 0;
        prefer[3] |= wave_ballot([&](int) -> int { return h3; }) != 0;
        const V64 kv = v64_make([&](int) -> int { return key; });
        const int kmin = imin(imin(v64_read(kv, 0), v64_read(kv, 16)), imin(v64_read(kv, 32), v64_read(kv, 48)));
        if (kmin != 0x7fffffff && (kmin >> 4) < sad_best + cand_cost_best)
        {
            mv_best = (mv32)cand.get(kmin & 15);
            cand_cost_best = mv_cost(m, mv_best, mv_pred16);
            sad_best = (kmin >> 4) - cand_cost_best;
        }
    }
    sad_best += mv_cost(m, mv_best, mv_pred16);
    {
        /* An upper bound of the cost this function will end with: the 16x16 search starts at sad_best and only improves it (diamond_g), the
         * decision is the minimum over the partition types, and when the skip vector's SAD is below that minimum the cost becomes
         * sad_skip + its vector cost (the last lines of this function), which may be larger than the minimum but not than this bound. */
        int u = MUL_LAMBDA(1, m.lambda_q4) + sad_best;
        if (sad_skip != 0x7FFFFFFF) u = imax(u, sad_skip + mv_cost(m, mv_skip, mv_pred16));
        sig.bound(u);
    }
    STAMP(L, 4);

    /* H:3646-3671: every partitioning is tried from the same predictor state; H:5283-5524's loop over the partition types runs as
     * four lane groups side by side (search_type).  The reference takes a type when its cost is strictly below the best so far, in the
     * order 16x16, 16x8, 8x16, 8x8 (H:5500): the minimum, the earliest type on ties. */
#ifdef H264E_ABLATE      /* what-if timing builds (tools/gpu_ablate.sh): NOT bit-exact, never the product */
    if (H264E_ABLATE == 1) prefer[3] = 0;
    if (H264E_ABLATE == 4) prefer[1] = prefer[2] = prefer[3] = 0;
#endif
    const int types = 1 | (prefer[1] ? 2 : 0) | (prefer[2] ? 4 : 0) | (prefer[3] ? 8 : 0);
#ifdef H264E_TYPES_PROBE
    PCOUNT(L, 20 + (types == 1 ? 0 : (types & 8) ? 2 : 1));
#endif
    WAVE_FOR(l)
    {
        const int i = l & 15;
        mv32 *c = &L.gctx[l >> 4].mv_left[0];              /* the 16 words of a GCtx: left[4], tl[4], top[8] */
        c[i] = i < 4 ? L.mv_left[i] : i < 8 ? L.mv_tl[i - 4] : i < 13 ? B.mv_top[i - 8] : 0;
    }
    wave_sync();
    /* three waves per row: the 8x8 type -- a chain of four searches, the long pole of this function -- runs on the helper wave, from the
     * same start vector and its own copy of the predictor context, while this wave searches the other types */
    const bool helped = sig.helper() && (types & 8);
    if (helped) sig.t3_request(mv_best, sad_best, lim);
    const int types_here = helped ? types & 7 : types;
    GRP_EACH(t)
    {
        if ((types_here >> t) & 1) search_type(L, B, m, t, mv_best, sad_best, lim, sad_skip != 0x7FFFFFFF ? sad_skip + mv_cost(m, mv_skip, mv_pred16) : 0);
    }
    wave_sync();
    if (helped && !sig.t3_wait()) { m.type = 0; m.cost = 0xffffff; B.mv[0] = 0; B.mvd[0] = 0; return; }      /* the row is being stopped: nothing of this macroblock is used */
    STAMP(L, 5);
    m.cost = 0xffffff;
    int best_n = 0;
    for (int t = 0; t < 4; t++)
        if ((types >> t) & 1)
        {
            const int c = uni(L.gcost[t]);
            if (c < m.cost) { m.cost = c; m.type = t; best_n = uni(L.gnum[t]); }
        }
    wave_copy_wh(B.pred, L.gtest[m.type], 16, 16);
    for (int i = 0; i < best_n; i++) { B.mv[i] = L.part_mv[m.type][i]; B.mvd[i] = L.part_mvd[m.type][i]; }
    wave_sync();

    if (m.cost > sad_skip)
    {
        m.type = 0;
        m.cost = sad_skip + mv_cost(m, mv_skip, mv_pred16);
        B.mv[0] = mv_skip;
        B.mvd[0] = mvsub(mv_skip, mv_pred16);
        wave_copy_wh(B.pred, L.skip_pred, 16, 16);
    }
}

/* ------------------------------------------------------------------ intra decisions */

/* H:4838-4858 intra_estimate_16x16 + H:4876-4896 intra_choose_16x16: picks the mode (m.i16_mode), leaves its prediction in L.i16pred and
 * returns its cost; the caller takes it when that is below the best cost so far (intra_merge) */
DEV int intra16_cost(RowLds &L, const MbBuf &B, MbCtx &m)
{
    const uint8_t *p = B.inp;
    const int v = m.avail & 3;             /* H:4840 mode_i16x16_valid: bit 0 = vertical allowed (top), bit 1 = horizontal (left) */
    int mode, sad4[4];
    int dx = iabs(p[0] - p[15]) + iabs(p[15*16] - p[15*16 + 15]) + iabs(p[8*16] - p[8*16 + 15]);
    int dy = iabs(p[0] - p[15*16]) + iabs(p[15] - p[15*16 + 15]) + iabs(p[8] - p[15*16 + 8]);
    if (dx > 30 + 3*dy && dy < (100 + 50 - m.qp) && (v & 1)) mode = 0;
    else if (dy > 30 + 3*dx && dx < (100 + 50 - m.qp) && (v & 2)) mode = 1;
    else mode = 2;
    m.i16_mode = mode;
    wave_pred16(L.i16pred, L.pix_left, B.pix_top, m.avail, mode);
    return wave_sad_lds_q(B.inp, L.i16pred, sad4) + MUL_LAMBDA(ue_len((uint32_t)mode + 1), m.lambda_q4) + m.lambda_i16;
}

#define I4_LOST 0x7fffffff
/*
 * H:4723-4833 intra_choose_4x4: 16 blocks in raster order, each predicted from reconstructed neighbours.  Returns the cost, or I4_LOST as
 * soon as the cost -- which only grows -- reaches `bound()`: intra 4x4 wins only with a cost strictly below the best of the other
 * decisions (H:4827), and nothing else of this function is observable for another macroblock type (mb_decide resets the mode contexts,
 * mb_write computes luma again).  bound() is asked again after every block: in the two-wave pipeline the inter cost may arrive while
 * this runs, and any upper bound of the final comparison value gives the same decision.  nz_mask_out: coded-block mask of a complete run.
 */
template <class BOUND> DEV int intra4_choose(RowLds &L, MbBuf &B, const MbCtx &m, BOUND bound, unsigned &nz_mask_out)
{
    /* H:4750-4752 block2avail {07 23 23 2b 9b 77 ff 77 9b ff ff 77 9b 77 ff 77}: low nibble = mask on the macroblock's
     * flags, high nibble = flags forced on; one byte per block, packed so the lookup needs no memory */
    const uint64_t b2a_lo = 0x77ff779b2b232307ull, b2a_hi = 0x77ff779b77ffff9bull;
    uint8_t *r0 = L.i4rec + 24 + 4;       /* sample (0,0); row stride 24, 4 spare columns on the left keep rows 4-byte aligned */
    const int avail = m.avail;
    int cost = m.lambda_i4;
    unsigned nz_mask = 0;
    if (cost >= bound()) return I4_LOST;
    const I4Lanes T = i4_lanes_make(24);                                            /* every lane's own table entries, for all 16 blocks */
    const I4Q K = i4q_make(L.qdat[0]);                                              /* ... and its quantiser / transform constants */
    WAVE_FOR(l)
    {
        if (l < 16) { r0[-24 + l] = B.pix_top[l]; r0[24*l - 1] = L.pix_left[l]; }
        else if (l < 20) r0[-24 + l] = B.pix_top[32 + l - 16];
        else if (l == 20) r0[-24 - 1] = L.pix_tl[0];
    }
    wave_sync();
    for (int n = 0; n < 16; n++)
    {
        const int r = n >> 2, c = n & 3;
        uint8_t *blk = r0 + 24*4*r + 4*c;
        const uint8_t *bin = B.inp + (c + r*16)*4;
        uint8_t *pr = L.tt;                                     /* prediction / reconstruction of this block, stride 16 */
        const int b2a = (int)(((n < 8 ? b2a_lo : b2a_hi) >> (8*(n & 7))) & 0xff);
        int a = (avail & b2a) | (b2a >> 4);
        if (!(b2a & AV_TL))
            if ((n <= 3 && (avail & AV_T)) || (n > 3 && (avail & AV_L))) a |= AV_TL;
        if (n < 3 && (avail & AV_T)) a |= AV_TR;
        int mpred = imin(L.i4_left[r], B.i4_top[c]);
        if (mpred < 0) mpred = 2;
        STAMP(L, 19);
        int res = wave_i4_choose(bin, pr, a, blk, mpred, MUL_LAMBDA(3, m.lambda_q4), L.i4s, T);
        STAMP(L, 31);
        const int mode = res & 15, sad = res >> 4;
        L.i4_left[r] = B.i4_top[c] = (int8_t)mode;
        L.i4_mode[n] = (int8_t)(mode == mpred ? -1 : mode > mpred ? mode - 1 : mode);
        unsigned coded = 0;
        cost += sad;
        if (cost >= bound()) return I4_LOST;
        if (sad > m.skip_thr_i4) coded = i4_block_code(K, bin, pr, blk, 24, L.qy + n);     /* transform, quantiser, reconstruction -> working picture */
        else
        {
            WAVE_FOR(l)
            {
                if (l < 16) { L.qy[n].qv[l] = 0; L.qy[n].dq[l] = 0; }
                if (l < 4) lds32_store(blk + 24*l, lds32(pr + 16*l));
            }
            wave_sync();
        }
        STAMP(L, 18);
        nz_mask = (nz_mask << 1) | coded;
    }
    nz_mask_out = nz_mask & 0xffff;
    return cost;
}

/* H:5748-5762: the intra candidates against the inter decision, in the reference's order (16x16 first, then 4x4), strict "<" */
DEV void intra_merge(RowLds &L, MbBuf &B, MbCtx &m, int cost16, int cost4, unsigned nz4)
{
    if (cost16 < m.cost)
    {
        m.cost = cost16;
        m.type = 6;
        wave_copy_wh(B.pred, L.i16pred, 16, 16);
    }
    if (cost4 != I4_LOST && cost4 < m.cost)
    {
        m.cost = cost4;
        m.type = 5;
        m.nz_mask = nz4;
    }
}

/* ------------------------------------------------------------------ decision -> contexts */

/*
 * The decision of the macroblock is final (m.type: -1 skip, 0..3 inter partitioning, 5 intra 4x4, 6 intra 16x16): bring the contexts
 * that the NEXT macroblock's decisions read up to date -- the vector predictor context (H:3696-3715 me_mv_medianpredictor_put as
 * mb_write calls it, H:4502, H:4560, H:4583: a skip and a 16x16 partition put the same vector, so the roll-back to skip inside mb_write
 * changes nothing here) and the intra 4x4 mode contexts (H:4404-4410) -- after keeping the neighbours' vectors for this macroblock's
 * deblocking strengths (H:5291-5296).  This is the hand-off point of the two-wave pipeline: the search of macroblock x + 1 starts here
 * while x is still being transformed, coded and filtered.
 */
DEV void mb_decide(RowLds &L, MbBuf &B, const MbCtx &m)
{
    if (m.slice_type == 0)
        for (int i = 0; i < 4; i++)
        {
            L.df_mv[4 + 5*i] = L.mv_left[i];
            L.df_mv[i] = B.mv_top[i];
        }
    wave_sync();
    if (m.type != 5)
        for (int i = 0; i < 4; i++) L.i4_left[i] = B.i4_top[i] = 2;
    if (m.type >= 5) mvp_put(L, B, 0, 0, 4, 4, MV_NA);
    else if (m.type <= 0) mvp_put(L, B, 0, 0, 4, 4, B.mv[0]);
    else
    {
        const int dx = (m.type & 2) ? 2 : 4, dyb = (m.type & 1) ? 2 : 4;
        int x = 0, y = 0;
        for (int part = 0;; part++)
        {
            mvp_put(L, B, x, y, dx, dyb, B.mv[part]);
            x = (x + dx) & 3;
            if (!x)
            {
                y = (y + dyb) & 3;
                if (!y) break;
            }
        }
    }
    wave_sync();
}

/* ------------------------------------------------------------------ macroblock write */

/* H:4378-4715 mb_write.  Reconstruction goes to the LDS deblock tiles; bits to the row buffer. */
DEV void mb_write(RowLds &L, MbBuf &B, MbCtx &m, BitW &b)
{
#define SCAN8(i) ((((i) >> 3) & 1)*8 + (((i) >> 1) & 1)*4 + (((i) >> 2) & 1)*2 + ((i) & 1))     /* H:920 decode_block_scan */
    const int i16 = m.type >= 6;
    int cbpl = 0, cbpc = 0, cbp = 0;
    uint8_t *nz = L.nzctx;               /* nz[9]: the diagonal nnz context of H:4385-4401, in LDS (indexed dynamically) */
    uint8_t *ty = L.ytile + 4*YT_STRIDE + 4;
    uint8_t *tc[2] = { L.ctile[0] + 2*CT_STRIDE + 2, L.ctile[1] + 2*CT_STRIDE + 2 };

    L.df_nzflag = ((L.df_nzflag >> 4) & 0x84210) | B.df_nz_top;
    for (int i = 0; i < 4; i++)
    {
        nz[5 + i] = B.nnz_top[i];
        nz[3 - i] = L.nnz_left[i];
    }
    nz[4] = 0;
    wave_sync();
    for (int i = 0; i < 4; i++) B.nnz_top[i] = L.nnz_left[i] = 0;

    if (m.type != -1)
    {
        PTIC();
        if (m.type != 5)
        {
            /* residual -> transform -> dead zone -> quantiser -> [luma DC] -> reconstruction in one register pass per 8x8 group (enc_kernels.h) */
            const unsigned mask = i16 ? wave_xform_quant_recon<QMODE_I16>(B.inp, B.pred, ty, YT_STRIDE, L.qy, L.dcy, L.lev_dcy, L.qdat[0], (int *)0)
                                      : wave_xform_quant_recon<QMODE_INTER>(B.inp, B.pred, ty, YT_STRIDE, L.qy, L.dcy, L.lev_dcy, L.qdat[0], (int *)0);
            m.nz_mask = mask & 0xffff;
        } else
        {
            WAVE_FOR(l)
            {
                int r = l >> 2, c = l & 3;
                lds32_store(ty + YT_STRIDE*r + 4*c, lds32(L.i4rec + 24 + 4 + 24*r + 4*c));
            }
            wave_sync();
        }
        PTOC(L, 30);
        if (m.nz_mask & 0xCC00) cbpl |= 1;
        if (m.nz_mask & 0x3300) cbpl |= 2;
        if (m.nz_mask & 0x00CC) cbpl |= 4;
        if (m.nz_mask & 0x0033) cbpl |= 8;

        for (int c = 0; c < 2; c++)
        {
            /* H:4453-4488: chroma AC with the dead zone, the 2x2 DC transform, reconstruction of every block once a DC level is coded (the
             * dequantised AC of a block without coded levels is zero in the registers: the reference zeroes its array for the same effect) */
            int dc_flag = 0;
            const unsigned mask = wave_xform_quant_recon<QMODE_CHROMA>(B.inp_c + 8*c, L.pred_c + 8*c, tc[c], CT_STRIDE, c ? L.qv : L.qu, c ? L.dcv : L.dcu,
                                                                       c ? L.lev_dcv : L.lev_dcu, L.qdat[1], &dc_flag);
            if (mask) cbpc = 2;
            cbpc |= dc_flag;
        }
        cbpc = imin(cbpc, 2);
        /* roll back to skip (H:4493-4499) */
        /* (uni: what comes back from LDS is a vector value to the compiler; a branch on it would make everything the branch touches -- the
         * bit writer's state first of all -- a vector value, and the whole syntax / CAVLC part would run on the vector unit) */
        if (!(m.type | cbpl | cbpc) && (mv32)uni((int)B.mv[0]) == m.mv_skip_pred) m.type = -1;
    }

    if (m.type == -1)
    {
        L.skip_run++;
        for (int i = 4; i < 8; i++) B.nnz_top[i] = L.nnz_left[i] = 0;
        for (int i = 0; i < 16; i++) L.df_mv[5 + 5*(i >> 2) + (i & 3)] = B.mv[0];
        WAVE_FOR(l)
        {
            int r = l >> 2, c = l & 3;
            lds32_store(ty + YT_STRIDE*r + 4*c, lds32(B.pred + 16*r + 4*c));
            if (l < 32)
            {
                int pl = l >> 4, rr = (l >> 1) & 7, g = l & 1;
                lds32_store(tc[pl] + CT_STRIDE*rr + 4*g, lds32(L.pred_c + 16*rr + 8*pl + 4*g));
            }
        }
        wave_sync();
    } else
    {
        int mb_type = m.type;
        if (i16)
        {
            if (cbpl) cbpl = 15;
            mb_type += m.i16_mode + cbpc*4 + (cbpl ? 12 : 0);
        }
        if (mb_type >= 5 && m.slice_type == 2) mb_type -= 5;
        if (m.slice_type != 2)
        {
            if (uni(L.coded_any)) bw_ue(b, (uint32_t)L.skip_run);
            else L.lead_skips = L.skip_run;             /* the finalizer writes this run: it may extend into earlier rows */
            L.skip_run = 0;
        }
        L.coded_any = 1;
        bw_ue(b, (uint32_t)mb_type);
        if (m.type == 3) bw_put(b, 4, 15);              /* four sub_mb_type ue(0) */
        if (m.type >= 5)
        {
            if (m.type == 5)
                for (int i = 0; i < 16; i++)
                {
                    const int md = uni((int)L.i4_mode[SCAN8(i)]);
                    if (md < 0) bw_put(b, 1, 1); else bw_put(b, 4, (uint32_t)md);
                }
            int cm = m.i16_mode;
            if (!(cm & 1)) cm ^= 2;
            bw_ue(b, (uint32_t)cm);
        } else
        {
            const int dx = (m.type & 2) ? 2 : 4, dyb = (m.type & 1) ? 2 : 4;
            int x = 0, y = 0;
            for (int part = 0;; part++)
            {
                bw_se(b, mvx(B.mvd[part]));
                bw_se(b, mvy(B.mvd[part]));
                for (int yy = 0; yy < dyb; yy++)
                    for (int xx = 0; xx < dx; xx++) L.df_mv[5 + 5*(y + yy) + x + xx] = B.mv[part];
                x = (x + dx) & 3;
                if (!x)
                {
                    y = (y + dyb) & 3;
                    if (!y) break;
                }
            }
        }
        cbp = cbpl + (cbpc << 4);
        if (!i16) bw_ue(b, k_cbp2code[m.type < 5][cbp]);
        if (cbp || i16) bw_se(b, 0);                    /* mb_qp_delta: QP is constant within a frame (no MB-level rate control) */
        if (i16) nz[4] = (uint8_t)cavlc_block(b, L.cavlc, L.lev_dcy, 0, 16, nz[3] + nz[5]);
        if (cbpl)
        {
            for (int i = 0; i < 16; i++)
            {
                const int j = SCAN8(i), k = 4 + (j & 3) - (j >> 2);
                if (cbp & (1 << (i >> 2)))
                {
                    nz[k] = (uint8_t)cavlc_block(b, L.cavlc, L.qy[j].qv, i16, 16 - i16, nz[k - 1] + nz[k + 1]);
                    if (nz[k]) L.df_nzflag |= 1u << (5 + (j & 3) + 5*(j >> 2));
                } else
                    nz[k] = 0;
            }
            for (int i = 0; i < 4; i++)
            {
                B.nnz_top[i] = nz[1 + i];
                L.nnz_left[i] = nz[7 - i];
            }
        }
        if (cbpc)
        {
            cavlc_block(b, L.cavlc, L.lev_dcu, 0, 4, 17 + 17);
            cavlc_block(b, L.cavlc, L.lev_dcv, 0, 4, 17 + 17);
            if (cbpc > 1)
                for (int c = 0; c < 2; c++)
                {
                    uint8_t *nzc = L.nzctx + 4;              /* nzc[5] aliases nz[4..8]: luma contexts are already stored back */
                    const int off = c ? 6 : 4;
                    const qblk_t *q = c ? L.qv : L.qu;
                    nzc[2] = 0;
                    for (int i = 0; i < 2; i++)
                    {
                        nzc[3 + i] = B.nnz_top[off + i];
                        nzc[1 - i] = L.nnz_left[off + i];
                    }
                    for (int i = 0; i < 4; i++)
                    {
                        const int k = 2 + (i & 1) - (i >> 1);
                        nzc[k] = (uint8_t)cavlc_block(b, L.cavlc, q[i].qv, 1, 15, nzc[k - 1] + nzc[k + 1]);
                    }
                    for (int i = 0; i < 2; i++)
                    {
                        B.nnz_top[off + i] = nzc[1 + i];
                        L.nnz_left[off + i] = nzc[3 - i];
                    }
                }
        }
        if (cbpc != 2)
            for (int i = 4; i < 8; i++) B.nnz_top[i] = L.nnz_left[i] = 0;
    }
    wave_sync();
#undef SCAN8
}

/* ------------------------------------------------------------------ deblock strengths */

/* H:5535-5637 df_strength + edge masking of H:5653-5661 -> L.bs */
DEV void df_strength(RowLds &L, const MbCtx &m, int top_type)
{
    /* one lane per 4x4 edge segment: lanes 0..15 the vertical edges (bs[4x + y]), 16..31 the horizontal ones (bs[16 + 4y + x]) */
    const uint32_t flag = L.df_nzflag;
    const int intra = m.type >= 5;
    const int strong_l = intra || (m.x && L.left_type >= 5), strong_t = intra || top_type >= 5;
    WAVE_FOR(l)
    {
        if (l < 32)
        {
            const int dir = l >> 4, i = l & 15;
            const int x = dir ? (i & 3) : (i >> 2), y = dir ? (i >> 2) : (i & 3), k = 5*y + x;
            int s;
            if (intra) s = i < 4 ? 0 : 3;
            else
            {
                const uint32_t f = flag >> k;
                const mv32 c = L.df_mv[k + 5], o = dir ? L.df_mv[k] : L.df_mv[k + 4];
                s = (f & (dir ? 33u : 0x30u)) ? 2 : (iabs(mvx(o) - mvx(c)) > 3 || iabs(mvy(o) - mvy(c)) > 3) ? 1 : 0;
            }
            if (i < 4)
            {
                if (dir ? strong_t : strong_l) s = 4;
                if (dir ? m.slice_top : !m.x) s = 0;          /* picture edge; h264-lab.h:5799-5803 no filtering across a slice's top edge */
            }
            L.bs[l] = (uint8_t)s;
        }
    }
    wave_sync();
}

#endif
