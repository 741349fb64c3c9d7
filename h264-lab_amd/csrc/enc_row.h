/*
 * enc_row.h -- row driver (row_begin / row_step / row_end) and the slice splice (splice_frame / finalize_commit), wave64 model.
 *
 * row_step = the reference's mb_encode (h264-lab.h:5724-5812) for macroblock (x, row): neighbour records of
 * the row above come from HBM (written by that row's wavefront), the left neighbour is in LDS.
 * splice_frame = the tail of encode_slice (h264-lab.h:6451-6456): concatenates the row bit buffers behind
 * the slice header, resolves mb_skip_run across rows, adds the RBSP trailing bits, and evaluates the
 * mv_clusters speculation.
 */
#ifndef H264E_ENC_ROW_H
#define H264E_ENC_ROW_H

#include "enc_mb.h"

#define GEOM_WIDE 0
#define GEOM_NARROW 1
#define GEOM_INTRA 2

/* the chain's buffers with explicit HBM address spaces (pointers loaded from a struct are generic otherwise: FLAT
 * instructions, and the compiler has to treat what they load as lane-varying) */
struct ChainG
{
    GLOBAL_AS h264e_mbbottom_t *bottom;
    GLOBAL_AS h264e_mbpend_t *pend;
    GLOBAL_AS int *progress;
    GLOBAL_AS uint32_t *rowbits;
    GLOBAL_AS h264e_rowmeta_t *rowmeta;
    GLOBAL_AS h264e_mbrec_t *mbrec;
    GLOBAL_AS uint8_t *arena;
    uint32_t arena_cap;
    GLOBAL_AS uint8_t *nal_arena;
    uint32_t nal_cap;
    GLOBAL_AS uint32_t *cursor;
    GLOBAL_AS h264e_frameout_t *fout;
    GLOBAL_AS int *far_reads;
    unsigned long long *prof;
};
DEV ChainG chain_view(const h264e_chain_dev_t &C)
{
    ChainG g;
    g.bottom = (GLOBAL_AS h264e_mbbottom_t *)C.bottom; g.pend = (GLOBAL_AS h264e_mbpend_t *)C.pend; g.progress = (GLOBAL_AS int *)C.progress;
    g.rowbits = (GLOBAL_AS uint32_t *)C.rowbits; g.rowmeta = (GLOBAL_AS h264e_rowmeta_t *)C.rowmeta;
    g.mbrec = (GLOBAL_AS h264e_mbrec_t *)C.mbrec; g.arena = (GLOBAL_AS uint8_t *)C.arena; g.arena_cap = C.arena_cap; g.nal_arena = (GLOBAL_AS uint8_t *)C.nal_arena; g.nal_cap = C.nal_cap;
    g.cursor = (GLOBAL_AS uint32_t *)C.cursor; g.fout = (GLOBAL_AS h264e_frameout_t *)C.fout; g.far_reads = (GLOBAL_AS int *)C.far_reads; g.prof = C.prof;
    return g;
}

DEV void row_begin(RowLds &L, const h264e_geom_t &G, const ChainG &C, const h264e_frame_task_t &T, int row)
{
    WAVE_FOR(l)
    {
        if (l < 4) { L.mv_left[l] = 0; L.mv_tl[l] = 0; L.i4_left[l] = -1; L.pix_tl[l] = 0; }
        if (l < 8) L.nnz_left[l] = NNZ_NA;
        if (l < 32) L.pix_left[l] = 0;
        if (l < 25) L.df_mv[l] = 0;
        L.strip_y[l] = 0;
        L.strip_c[l >> 5][l & 31] = 0;
    }
    L.df_nzflag = 0;
    L.left_type = 0;
    L.left_qp = T.qp;
    L.bw.acc = 0; L.bw.nacc = 0; L.bw.pos = 0; L.bw.overflow = 0;
    L.bw.cap = (uint32_t)G.row_words;
    L.bw.buf = C.rowbits + (size_t)row*G.row_words;
    L.skip_run = 0; L.lead_skips = 0; L.coded_any = 0; L.far_reads[0] = L.far_reads[1] = L.far_reads[2] = 0; L.far_fail[0] = L.far_fail[1] = L.far_fail[2] = 0;
    L.f_noskip = L.f_bound = L.f_inter = L.f_decided = L.f_wdone = L.f_stop = 0; L.early_bound = 0; L.f_t3req = L.f_t3done = L.f_front = 0;
    WAVE_FOR(l) { if (l <= H264E_MAX_SLICES) L.slice_row[l] = l <= T.nslices ? T.slice_row[l] : (int16_t)0x7fff; }
    WAVE_FOR(l)
    {
        for (int k = l; k < 84; k += 64) L.qdat[k/42][k%42] = T.qdat[k/42][k%42];
    }
    cavlc_tab_load(L.cavlc);
    df_tab_load(L.dftab);
    L.qconst[0] = k_lambda_mv_q4[T.qp]; L.qconst[1] = k_lambda_q4[T.qp]; L.qconst[2] = k_skip_thr_inter[T.qp];
    L.qconst[3] = k_skip_thr_i4x4[T.qp]; L.qconst[4] = k_lambda_i4_q4[T.qp]; L.qconst[5] = k_lambda_i16_q4[T.qp];
    for (int i = 0; i < 32; i++) L.prof[0][i] = L.prof[1][i] = L.prof[2][i] = L.prof[3][i] = 0;
    L.prof_last[0] = L.prof_last[1] = 0;
    PROF_ROW_BEGIN(L);
    wave_sync();
    PROF_ROW_SYNC(L);
}

/* neighbour record of the macroblock above -> LDS (h264-lab.h:742-745 contexts, :590-606 deblock state): 16 coherent dword
 * loads of the record (+2 of the record to its right), staged in LDS, then unpacked */
/* part: LOAD_TOP_ALL, or -- two waves per row -- LOAD_TOP_MV for the search wave (the vectors of the macroblock above and of the one to its
 * right: all the search needs, and available as soon as the row above has DECIDED them, h264e_kernels.hip) and LOAD_TOP_REST for the
 * reconstruction wave (samples, contexts, pending lines: what the row above writes at the end of its macroblock) */
#define LOAD_TOP_ALL 0
#define LOAD_TOP_MV 1
#define LOAD_TOP_REST 2
template <int PART> DEV void load_top(RowLds &L, MbBuf &B, const h264e_geom_t &G, const GLOBAL_AS h264e_mbbottom_t *above, const GLOBAL_AS h264e_mbpend_t *pend_above, int x, int have_top)
{
    if (PART == LOAD_TOP_MV)
    {
        WAVE_FOR(l)
        {
            if (l < 4) B.mv_top[l] = have_top ? (mv32)cload32((const gu8 *)(above + x) + 32 + 4*l) : 0;
            else if (l == 4) B.mv_top[4] = (have_top && x + 1 < G.nmbx) ? (mv32)cload32((const gu8 *)(above + x + 1) + 32) : 0;
        }
        wave_sync();
        return;
    }
    if (have_top)
    {
        WAVE_FOR(l)
        {
            if (l < 16) lds32_store(L.trec + 4*l, cload32((const gu8 *)(above + x) + 4*l));
            else if (l == 16) lds32_store(L.trec + 64, x + 1 < G.nmbx ? cload32((const gu8 *)(above + x + 1)) : 0u);                /* pix[0..3] of the above-right record */
            else if (l == 17) lds32_store(L.trec + 68, x + 1 < G.nmbx ? cload32((const gu8 *)(above + x + 1) + 32) : 0u);           /* its mv[0] */
            else if (l >= 32 && l < 56) lds32_store(B.ptop + 4*(l - 32), cload32((const gu8 *)(pend_above + x) + 4*(l - 32)));    /* for the deblocking at the end of the step */
        }
        wave_sync();
        WAVE_FOR(l)
        {
            if (l < 8) lds32_store(B.pix_top + 4*l, lds32(L.trec + 4*l));
            else if (l < 12) { if (PART == LOAD_TOP_ALL) B.mv_top[l - 8] = (mv32)lds32(L.trec + 32 + 4*(l - 8)); }
            else if (l < 20) B.nnz_top[l - 12] = L.trec[48 + l - 12];
            else if (l < 24) B.i4_top[l - 20] = (int8_t)L.trec[56 + l - 20];
            else if (l == 24) { B.df_nz_top = L.trec[60]; B.top_type = (int8_t)L.trec[61]; B.top_qp = L.trec[62]; }
            else if (l == 25) lds32_store(B.pix_top + 32, lds32(L.trec + 64));
            else if (l == 26) { if (PART == LOAD_TOP_ALL) B.mv_top[4] = (mv32)lds32(L.trec + 68); }
        }
    } else
    {
        WAVE_FOR(l)
        {
            if (l < 36) B.pix_top[l] = 0;
            if (l < 5 && PART == LOAD_TOP_ALL) B.mv_top[l] = 0;
            if (l < 8) B.nnz_top[l] = NNZ_NA;
            if (l < 4) B.i4_top[l] = -1;
            if (l == 0) { B.df_nz_top = 0; B.top_type = 0; B.top_qp = 0; }
        }
    }
    wave_sync();
}

/* h264-lab.h:5731-5740 + 3536-3562: input macroblock -> LDS, replicating the last valid column / row of cropped pictures */
DEV void load_input(MbBuf &B, const h264e_geom_t &G, const RowTask &T, int mbx, int mby)
{
    /* a macroblock that lies inside the picture (all but the last column / row of a cropped picture) is 96 dword loads; only the
     * cropped edge replicates sample by sample */
    const bool inside = (mbx + 1)*16 <= G.width && (mby + 1)*16 <= G.height;
    WAVE_FOR(l)
    {
        {
            int r = l >> 2, c = l & 3, yy = imin(mby*16 + r, G.height - 1);
            const gu8 *p = (const gu8 *)T.in[0] + (size_t)yy*T.in_stride[0];
            uint32_t v = 0;
            if (inside) v = gload32(p + mbx*16 + 4*c);
            else for (int k = 0; k < 4; k++) v |= (uint32_t)p[imin(mbx*16 + 4*c + k, G.width - 1)] << (8*k);
            lds32_store(B.inp + 16*r + 4*c, v);
        }
        if (l < 32)
        {
            int pl = l >> 4, r = (l >> 1) & 7, c = l & 1, yy = imin(mby*8 + r, G.height/2 - 1);
            /* (selected, not indexed: a lane-varying index would put the task copy into scratch memory) */
            const gu8 *p = (const gu8 *)(pl ? T.in[2] : T.in[1]) + (size_t)yy*(pl ? T.in_stride[2] : T.in_stride[1]);
            uint32_t v = 0;
            if (inside) v = gload32(p + mbx*8 + 4*c);
            else for (int k = 0; k < 4; k++) v |= (uint32_t)p[imin(mbx*8 + 4*c + k, G.width/2 - 1)] << (8*k);
            lds32_store(B.inp_c + 16*r + 8*pl + 4*c, v);
        }
    }
    wave_sync();
}

/* What macroblock (x, row) needs that does not depend on the row above: its input samples and, for P slices, the
 * reference window (the caller has waited for the temporal dependency).  Issued BEFORE the wait for the row above, so
 * the HBM latency of these loads overlaps with that wait. */
template <int GEOM> DEV void row_prefetch(RowLds &L, const h264e_geom_t &G, const RowTask &T, int row, int x)
{
    load_input(L.mb[x & 1], G, T, x, row);
    if (GEOM != GEOM_INTRA && T.slice_type == 0)
    {
        Plane P;
        P.p = (const gu8 *)T.ref[0]; P.w = G.W; P.h = G.H; P.stride = G.W;
        wave_load_window(L.win, P, x*16 - WIN_M, row*16 - WIN_M, T.narrow);
    }
}

/*
 * One macroblock = three phases (the reference's mb_encode, h264-lab.h:5724-5812, cut where its data flow allows two instruction streams):
 *   mb_search        neighbour records of the row above -> LDS, inter decision (H:5283-5524)                     [search side]
 *   mb_intra_decide  intra 16x16 / 4x4 candidates (H:5748-5762), the final decision, contexts for the next one   [reconstruction side]
 *   mb_recon_write   chroma prediction, mb_write (H:4378-4715), deblocking (H:5535-5716), stores                 [reconstruction side]
 * The search of macroblock x + 1 needs nothing of x but the predictor context that mb_intra_decide leaves behind, and the intra
 * candidates of x + 1 need the reconstruction of x (left column) -- so with two wavefronts per row (h264e_kernels.hip) the search
 * wave works on x + 1 while the reconstruction wave writes x, then tests the intra candidates of x + 1.  With one wavefront (and in
 * the emulation) the three run one after the other; the decisions are the same either way.
 * GEOM: compile-time kernel variant -- GEOM_WIDE / GEOM_NARROW = the reference-window geometry (h264e_dev.h); GEOM_INTRA = a launch of
 * intra frames only: the inter decision, the reference window and the inter chroma prediction are compiled out (and with them most of
 * the register pressure: that variant runs at twice the waves per SIMD).
 * row0 / row1: first row and end row of the slice (row band) this row belongs to -- the whole picture for one slice per frame.
 * A slice is encoded like a picture of its own as far as neighbour availability, contexts and deblocking are concerned
 * (h264-lab.h:3605-3622 mb_avail_flag relative to slice.start_mb_num, h264-lab.h:5799-5808 no filtering across its top edge).
 */
template <int GEOM> DEV void mb_ctx_init(MbCtx &m, RowLds &L, const h264e_geom_t &G, const RowTask &T, int row, int x, int row0, int side)
{
    const bool have_top = row > row0;
    m.G = &G;
    m.speed = T.speed; m.slice_type = T.slice_type; m.clu[0] = T.clusters[0]; m.clu[1] = T.clusters[1]; m.clu_per_mb = T.clusters_per_mb;
    for (int c = 0; c < 3; c++)
    {
        m.ref[c].p = (const gu8 *)T.ref[c];
        m.ref[c].w = G.W >> (c ? 1 : 0); m.ref[c].h = G.H >> (c ? 1 : 0); m.ref[c].stride = m.ref[c].w;
        m.dec[c] = (gu8 *)T.dec[c];
    }
    m.x = x; m.y = row; m.num = row*G.nmbx + x;
    m.avail = (have_top ? AV_T : 0) | (have_top && x != G.nmbx - 1 ? AV_TR : 0) | (x > 0 ? AV_L : 0) | (have_top && x > 0 ? AV_TL : 0);
    m.slice_top = !have_top;
    m.cropped = G.cropping && ((x + 1)*16 > G.width || (row + 1)*16 > G.height);
    m.type = 0; m.cost = 0x7FFFFFFF; m.i16_mode = 0; m.used_cand = 0; m.mv_skip_pred = 0; m.nz_mask = 0;
    m.qp = T.qp;
    m.lambda_mv = uni(L.qconst[0]); m.lambda_q4 = uni(L.qconst[1]); m.skip_thr = uni(L.qconst[2]);
    m.skip_thr_i4 = uni(L.qconst[3]); m.lambda_i4 = uni(L.qconst[4]); m.lambda_i16 = uni(L.qconst[5]);
    m.rv.dep = (const GLOBAL_AS int *)T.dep_progress; m.rv.nmbx = G.nmbx; m.rv.nmby = G.nmby;
    m.rv.P = m.ref[0]; m.rv.win = (const lu8 *)L.win; m.rv.has_win = GEOM != GEOM_INTRA && T.slice_type == 0; m.rv.wx0 = x*16 - WIN_M; m.rv.wy0 = row*16 - WIN_M;
    m.rv.vw = GEOM == GEOM_NARROW ? H264E_NARROW_VW : WIN_W; m.rv.vh = GEOM == GEOM_NARROW ? H264E_NARROW_VH : WIN_W; m.rv.far = &L.far_reads[side]; m.rv.fail = &L.far_fail[side]; m.rv.slice_row = L.slice_row; m.rv.nslices = T.nslices; m.rv.spin_limit = G.spin_limit;
}

/* search side.  The input macroblock, the reference window (row_prefetch) and the records of the row above (load_top) are already in
 * LDS.  sig (inter_choose): sig.noskip() is called once the
 * early-skip test has failed (the reconstruction side may start on the intra candidates then). */
struct NoSignals { DEVM void noskip() const {} DEVM void bound(int) const {} DEVM bool helper() const { return false; } DEVM void t3_request(mv32, int, const rect_t &) const {} DEVM bool t3_wait() const { return true; } };
template <int GEOM, class SIG> DEV void mb_search(RowLds &L, MbBuf &B, const h264e_geom_t &G, const RowTask &T, int row, int x, int row0, SIG sig)
{
    MbCtx m;
    mb_ctx_init<GEOM>(m, L, G, T, row, x, row0, 0);
    STAMP(L, 1);
    if (GEOM != GEOM_INTRA && T.slice_type == 0) inter_choose(L, B, m, sig);
    STAMP(L, 7);
    B.type = m.type; B.cost = m.cost; B.used_cand = m.used_cand; B.mv_skip_pred = m.mv_skip_pred;
    wave_sync();
}

/* How the reconstruction side learns about the inter decision: it is simply there (one wavefront, emulation) ... */
struct InterIsThere
{
    DEVM bool ready() const { return true; }
    DEVM bool wait_noskip_or_ready() const { return true; }
    DEVM bool wait_bound_or_ready() const { return true; }
    DEVM bool wait_ready() const { return true; }
    DEVM int early_bound() const { return 0x7fffffff; }
    DEVM bool before_decide() const { return true; }
};

/* reconstruction side, first half: the intra candidates and the final decision.  The intra candidates do not need the inter decision
 * -- only the comparison at the end does -- so with a search wave still at work (P.ready() false) they start as soon as the early-skip
 * test has failed, bounded by the 16x16 intra cost until the inter cost arrives.  Returns false when the row has to stop. */
template <class P> DEV bool mb_intra_decide(RowLds &L, MbBuf &B, MbCtx &m, const RowTask &T, P pol)
{
    bool have = pol.ready();
    if (!have)
    {
        if (!pol.wait_noskip_or_ready()) return false;
        have = pol.ready();
    }
    STAMP(L, 15);               /* two waves: waited for the search wave's early-skip test */
    if (have) { m.type = uni(B.type); m.cost = uni(B.cost); }
    if (m.type >= 0)
    {
        const int cost16 = intra16_cost(L, B, m);
        STAMP(L, 8);
        int cost4 = I4_LOST, bnd = imin(m.cost, cost16);
        unsigned nz4 = 0;
        /* the 4x4 candidates cost three times the 16x16 one and lose against most inter decisions after a few blocks -- once there is a
         * bound to lose against: wait for the search wave's first one (it follows the early-skip test by one candidate scan) */
        if (!have && !pol.wait_bound_or_ready()) return false;
#ifdef H264E_ABLATE
        if (H264E_ABLATE != 2)
#endif
        if (T.speed < 2 || T.slice_type != 0)
            cost4 = intra4_choose(L, B, m, [&]() -> int {
                if (!have)
                {
                    if (pol.ready()) { have = true; bnd = imin(bnd, uni(B.cost)); }
                    else bnd = imin(bnd, pol.early_bound());         /* the search wave's upper bound of the inter cost, once it has one */
                }
                return bnd;
            }, nz4);
        STAMP(L, 9);
        if (!pol.wait_ready()) return false;
        STAMP(L, 16);           /* two waves: waited for the inter decision */
        m.type = uni(B.type); m.cost = uni(B.cost);
        if (!pol.before_decide()) return false;
        if (m.type >= 0) intra_merge(L, B, m, cost16, cost4, nz4);
    } else if (!pol.before_decide()) return false;
    m.used_cand = uni(B.used_cand); m.mv_skip_pred = (mv32)uni(B.mv_skip_pred);
    mb_decide(L, B, m);
    return true;
}

/* reconstruction side, second half */
struct NoHook { DEVM void operator()() const {} };
/* ... in two parts, because a fourth wave can take the second one (h264e_kernels.hip, the latency variant): mb_recon_front = chroma
 * prediction, mb_write, the validation record, the unfiltered edges the next macroblock's intra prediction needs; mb_recon_back =
 * deblocking and every store of the macroblock (picture, pending lines, the record for the row below) -- nothing the front part of the
 * NEXT macroblock needs before its mb_decide */
template <int GEOM, class HOOK> DEV void mb_recon_front(RowLds &L, MbBuf &B, MbCtx &m, const h264e_geom_t &G, const ChainG &C, const RowTask &T, int row, int x, int row0, int row1, HOOK after_prediction)
{
    if (GEOM == GEOM_INTRA || m.type >= 5) wave_pred_chroma(L.pred_c, L.pix_left + 16, B.pix_top + 16, m.avail, m.i16_mode);
    else predict_chroma_inter(B, m, L.pred_c);
    after_prediction();             /* two waves per row: the `decided` counter (h264e_kernels.hip) */

    STAMP(L, 10);
    BitW bw = bw_uniform(L.bw);
    mb_write(L, B, m, bw);
    L.bw = bw;
    STAMP(L, 11);

    /* record for the mv_clusters validation (h264-lab.h:5776-5779 updates them with mv[0] of every non-intra MB) */
    {
        GLOBAL_AS h264e_mbrec_t *rec = C.mbrec + (size_t)T.frame_slot*G.nmb + m.num;
        const uint32_t mv0 = m.type < 5 ? (uint32_t)B.mv[0] : 0u;
        const uint64_t w = (uint64_t)mv0 | ((uint64_t)(uint8_t)(int8_t)m.type << 32) | ((uint64_t)(m.used_cand & 255) << 40);
        cstore64((gu8 *)rec, w);                /* (every lane the same word: no divergent branch, see enc_kernels.h bw_put) */
    }

    /* keep the UNFILTERED right column / bottom row for intra prediction (h264-lab.h:4693-4714) */
    uint8_t *ty = L.ytile + 4*YT_STRIDE + 4;
    uint8_t *tc0 = L.ctile[0] + 2*CT_STRIDE + 2, *tc1 = L.ctile[1] + 2*CT_STRIDE + 2;
    /* the record for the row below is assembled in LDS (L.brec) and stored at the end of the step */
    WAVE_FOR(l)
    {
        if (l < 16)
        {
            L.pix_left[l] = ty[YT_STRIDE*l + 15];
            L.brec[l] = ty[YT_STRIDE*15 + l];
        } else if (l < 32)
        {
            int pl = (l >> 3) & 1, i = l & 7;
            const uint8_t *t = pl ? tc1 : tc0;
            L.pix_left[16 + 8*pl + i] = t[CT_STRIDE*i + 7];
            L.brec[16 + 8*pl + i] = t[CT_STRIDE*7 + i];
        } else if (l < 35)
        {
            int c = l - 32;
            L.pix_tl[c] = B.pix_top[c == 0 ? 15 : 15 + 8*c];
        }
    }
    wave_sync();

}

template <int GEOM> DEV void mb_recon_back(RowLds &L, MbBuf &B, MbCtx &m, const h264e_geom_t &G, const ChainG &C, const RowTask &T, int row, int x, int row0, int row1)
{
    const bool have_top = row > row0;
    GLOBAL_AS h264e_mbbottom_t *rowrec = C.bottom + (size_t)row*G.nmbx;
    uint8_t *ty = L.ytile + 4*YT_STRIDE + 4;
    uint8_t *tc0 = L.ctile[0] + 2*CT_STRIDE + 2, *tc1 = L.ctile[1] + 2*CT_STRIDE + 2;
    /* deblock on the LDS tiles: left strips from LDS (previous macroblock), top strips from the pending lines of the
     * row above (h264e_mbpend_t) */
    const int W = G.W, Wc = G.W >> 1;
    gu8 *dy = m.dec[0] + (size_t)(row*16)*W + x*16;
    gu8 *du = m.dec[1] + (size_t)(row*8)*Wc + x*8, *dv = m.dec[2] + (size_t)(row*8)*Wc + x*8;
    GLOBAL_AS h264e_mbpend_t *pend_row = C.pend + (size_t)row*G.nmbx;
    const bool direct = T.no_deblock || row == row1 - 1;        /* nothing below will filter the bottom lines (picture or slice end): they are final now */
    if (!T.no_deblock)
    {
        df_strength(L, m, B.top_type);
        WAVE_FOR(l)
        {
            if (l < 16) lds32_store(L.ytile + (4 + l)*YT_STRIDE, lds32(L.strip_y + 4*l));
            else if (l < 32)
            {
                int pl = (l >> 3) & 1, i = l & 7;
                uint8_t *t = L.ctile[pl] + (2 + i)*CT_STRIDE;
                t[0] = L.strip_c[pl][4*i + 2]; t[1] = L.strip_c[pl][4*i + 3];
            } else if (l < 48)
            {
                int r = (l - 32) >> 2, c = l & 3;
                uint32_t v = 0;
                if (have_top) v = lds32(B.ptop + 16*r + 4*c);
                lds32_store(L.ytile + r*YT_STRIDE + 4 + 4*c, v);
            } else if (l < 56)
            {
                int pl = (l >> 2) & 1, r = (l >> 1) & 1, c = l & 1;
                uint32_t v = 0;
                if (have_top) v = lds32(B.ptop + 64 + 16*pl + 8*r + 4*c);
                memcpy(L.ctile[pl] + r*CT_STRIDE + 2 + 4*c, &v, 4);
            }
        }
        wave_sync();
        wave_deblock(L.ytile, L.ctile[0], L.ctile[1], L.bs, T.qp, L.left_qp, B.top_qp, L.dftab);
    }
    /* write the macroblock: final lines into the picture, the bottom lines into the pending record; 8 bytes per lane (every
     * store is one write-through fabric request, whatever its width) */
    WAVE_FOR(l)
    {
        if (l < 32)
        {
            const int r = l >> 1, hf = l & 1;
            const uint64_t v = (uint64_t)lds32(ty + YT_STRIDE*r + 8*hf) | ((uint64_t)lds32(ty + YT_STRIDE*r + 8*hf + 4) << 32);
            if (r < 12 || direct) cstore64(dy + (size_t)r*W + 8*hf, v);
            else cstore64((gu8 *)pend_row[x].y + 16*(r - 12) + 8*hf, v);
        } else if (l < 48)
        {
            const int pl = (l - 32) >> 3, rr = l & 7;
            const uint8_t *t = (pl ? tc1 : tc0) + CT_STRIDE*rr;
            const uint64_t u = (uint64_t)lds32(t) | ((uint64_t)lds32(t + 4) << 32);
            if (rr < 6 || direct) cstore64((pl ? dv : du) + (size_t)rr*Wc, u);
            else cstore64((gu8 *)pend_row[x].c[pl] + 8*(rr - 6), u);
        }
    }
    if (!T.no_deblock)
    {
        /* the neighbour samples the filter changed: 3 luma / 1 chroma columns of the left macroblock (picture, or its pending
         * record for the bottom lines) -- unless the left edge was not filtered at all (strength 0 on all of it: two skipped
         * macroblocks side by side), then what the left macroblock stored itself stands -- and the 4 luma / 2 chroma lines above,
         * which are final now */
        const bool left_filtered = x > 0 && uni(lds32(L.bs)) != 0;
        WAVE_FOR(l)
        {
            if (l < 16)
            {
                if (left_filtered)
                {
                    uint32_t v = lds32(L.ytile + (4 + l)*YT_STRIDE);
                    if (l < 12 || direct) cstore32(dy + (size_t)l*W - 4, v);
                    else cstore32((gu8 *)pend_row[x - 1].y + 16*(l - 12) + 12, v);
                }
            } else if (l < 24)
            {
                int r = (l - 16) >> 1, hf = l & 1;                          /* tile rows 0..3 = picture rows -4..-1 */
                if (have_top)
                {
                    const uint8_t *t = L.ytile + r*YT_STRIDE + 4 + 8*hf;
                    cstore64(dy - (size_t)(4 - r)*W + 8*hf, (uint64_t)lds32(t) | ((uint64_t)lds32(t + 4) << 32));
                }
            } else if (l < 32)
            {
            } else if (l < 48)
            {
                int pl = (l - 32) >> 3, i = (l - 32) & 7;
                if (left_filtered)
                {
                    /* columns 4..7 of the left macroblock as one dword: 4..6 as it left them, 7 as this filter left it */
                    const uint32_t v = (uint32_t)L.strip_c[pl][4*i] | ((uint32_t)L.strip_c[pl][4*i + 1] << 8) | ((uint32_t)L.strip_c[pl][4*i + 2] << 16) |
                                       ((uint32_t)L.ctile[pl][(2 + i)*CT_STRIDE + 1] << 24);
                    if (i < 6 || direct) cstore32((pl ? dv : du) + (size_t)i*Wc - 4, v);
                    else cstore32((gu8 *)pend_row[x - 1].c[pl] + 8*(i - 6) + 4, v);
                }
            } else if (l < 52)
            {
                int pl = (l - 48) >> 1, r = l & 1;                          /* tile rows 0..1 = picture rows -2..-1 */
                if (have_top)
                {
                    uint32_t lo, hi;
                    memcpy(&lo, L.ctile[pl] + r*CT_STRIDE + 2, 4); memcpy(&hi, L.ctile[pl] + r*CT_STRIDE + 6, 4);
                    cstore64((pl ? dv : du) - (size_t)(2 - r)*Wc, (uint64_t)lo | ((uint64_t)hi << 32));
                }
            }
        }
    }
    /* carried deblock state: deblocked right columns of this macroblock; rest of the record for the row below */
    WAVE_FOR(l)
    {
        if (l < 16) lds32_store(L.strip_y + 4*l, lds32(ty + YT_STRIDE*l + 12));
        else if (l < 32)
        {
            int pl = (l >> 3) & 1, i = l & 7;
            const uint8_t *t = (pl ? tc1 : tc0) + CT_STRIDE*i + 4;
            L.strip_c[pl][4*i] = t[0]; L.strip_c[pl][4*i + 1] = t[1]; L.strip_c[pl][4*i + 2] = t[2]; L.strip_c[pl][4*i + 3] = t[3];
        } else if (l < 36) lds32_store(L.brec + 32 + 4*(l - 32), (uint32_t)B.mv_top[l - 32]);
        else if (l < 44) L.brec[48 + l - 36] = B.nnz_top[l - 36];
        else if (l < 48) L.brec[56 + l - 44] = (uint8_t)B.i4_top[l - 44];
        else if (l == 48)
        {
            L.brec[60] = (uint8_t)(L.df_nzflag >> 20);
            L.brec[61] = (uint8_t)(int8_t)m.type;
            L.brec[62] = (uint8_t)T.qp;
            L.brec[63] = 0;
        }
    }
    wave_sync();
    WAVE_FOR(l) { if (l < 16) cstore32((gu8 *)(rowrec + x) + 4*l, lds32(L.brec + 4*l)); }
    L.left_type = m.type;
    L.left_qp = T.qp;
    wave_sync();
    STAMP(L, 12);
#ifndef H264E_TYPES_PROBE
    PCOUNT(L, 20 + (m.type < 0 ? 0 : m.type < 5 ? 1 : 2));
#endif
}

template <int GEOM, class HOOK> DEV void mb_recon_write(RowLds &L, MbBuf &B, MbCtx &m, const h264e_geom_t &G, const ChainG &C, const RowTask &T, int row, int x, int row0, int row1, HOOK after_prediction)
{
    mb_recon_front<GEOM>(L, B, m, G, C, T, row, x, row0, row1, after_prediction);
    mb_recon_back<GEOM>(L, B, m, G, C, T, row, x, row0, row1);
}

/* the three phases one after the other: one wavefront per row, and the emulation */
template <int GEOM> DEV void row_step(RowLds &L, const h264e_geom_t &G, const ChainG &C, const RowTask &T, int row, int x, int row0, int row1)
{
    MbBuf &B = L.mb[x & 1];
    MbCtx m;
    load_top<LOAD_TOP_ALL>(L, B, G, C.bottom + (size_t)(row - 1)*G.nmbx, C.pend + (size_t)(row - 1)*G.nmbx, x, row > row0);
    mb_search<GEOM>(L, B, G, T, row, x, row0, NoSignals());
    mb_ctx_init<GEOM>(m, L, G, T, row, x, row0, 1);
    mb_intra_decide(L, B, m, T, InterIsThere());
    mb_recon_write<GEOM>(L, B, m, G, C, T, row, x, row0, row1, NoHook{});
}

DEV void row_end(RowLds &L, const h264e_geom_t &G, const ChainG &C, int row)
{
    BitW bw = bw_uniform(L.bw);
    const uint32_t nbits = bw_bits(bw);
    if (bw.nacc)
    {
        const int pad = 32 - bw.nacc;
        bw_put(bw, pad, 0);
    }
    gu8 *M = (gu8 *)(C.rowmeta + row);       /* {nbits, lead_skips, trail_skips, overflow}: read by the finalizer workgroup */
    if (wave_lane() == 0)
    {
        g_atomic_add(C.far_reads, L.far_reads[0] + L.far_reads[1] + L.far_reads[2]);
        cstore32(M, nbits);
        cstore32(M + 4, (uint32_t)(L.coded_any ? L.lead_skips : G.nmbx));
        cstore32(M + 8, (uint32_t)(L.coded_any ? L.skip_run : 0));
        cstore32(M + 12, (uint32_t)bw.overflow);
    }
    wave_sync();
    PROF_ROW_END(L, C);
}

/* ------------------------------------------------------------------ slice splice (one wavefront per chain) */

struct SpliceState { GLOBAL_AS uint32_t *out; uint32_t wpos; uint32_t carry; int cbits; uint32_t cap_words; int overflow; };

/* append nbits (<= 64) right-aligned bits */
DEV void splice_put(SpliceState &s, int nbits, uint64_t v)
{
    while (nbits > 0)
    {
        int take = imin(nbits, 32 - s.cbits);
        uint32_t part = (uint32_t)((v >> (nbits - take)) & ((take == 32) ? 0xffffffffu : ((1u << take) - 1)));
        s.carry = (take == 32) ? part : ((s.carry << take) | part);
        s.cbits += take;
        nbits -= take;
        if (s.cbits == 32)
        {
            if (s.wpos < s.cap_words) s.out[s.wpos] = bswap32(s.carry); else s.overflow = 1;
            s.wpos++;
            s.carry = 0; s.cbits = 0;
        }
    }
}

/* append the first nbits of an MSB-first word buffer, 64 words per pass */
DEV void splice_words(SpliceState &s, const GLOBAL_AS uint32_t *w, uint32_t nbits)
{
    const uint32_t nfull = nbits >> 5;
    const int cb = s.cbits;
    const uint32_t carry = s.carry;
    for (uint32_t base = 0; base < nfull; base += 64)
    {
        WAVE_FOR(l)
        {
            uint32_t k = base + (uint32_t)l;
            if (k < nfull)
            {
                uint32_t prev = k ? w[k - 1] : carry, cur = w[k];
                uint32_t o = cb ? ((prev << (32 - cb)) | (cur >> cb)) : cur;
                if (s.wpos + k < s.cap_words) s.out[s.wpos + k] = bswap32(o);
            }
        }
    }
    if (s.wpos + nfull > s.cap_words) s.overflow = 1;
    if (nfull)
    {
        s.wpos += nfull;
        s.carry = cb ? (w[nfull - 1] & ((1u << cb) - 1)) : 0;
    }
    const int rem = (int)(nbits & 31);
    if (rem) splice_put(s, rem, (uint64_t)(w[nfull] >> (32 - rem)));
}

DEV void put_ue64(SpliceState &s, uint32_t v)
{
    int n = 2*(32 - clz32(v + 1)) - 1;
    splice_put(s, n, (uint64_t)v + 1);
}

DEV void clusters_step(mv32 c[2], mv32 mv)                                  /* h264-lab.h:5263-5278 */
{
    int n = mvx(mv)*mvx(mv) + mvy(mv)*mvy(mv);
    int n0 = mvx(c[0])*mvx(c[0]) + mvy(c[0])*mvy(c[0]), n1 = mvx(c[1])*mvx(c[1]) + mvy(c[1])*mvy(c[1]);
    if (n < n1) c[0] = mvmk((63*mvx(c[0]) + mvx(mv) + 32) >> 6, (63*mvy(c[0]) + mvy(mv) + 32) >> 6);
    if (n >= n0) c[1] = mvmk((63*mvx(c[1]) + mvx(mv) + 32) >> 6, (63*mvy(c[1]) + mvy(mv) + 32) >> 6);
}

/*
 * The exact mv_clusters walk of one frame on the device (what h264e_host.c clusters_walk does on the host): from state s, in
 * raster order -- restarting from s at every slice of a multi-slice frame -- compare the rounded candidates every macroblock
 * CONSUMED (the frame-constant pair or its entry of the per-macroblock array) with the exact ones, then apply its update
 * (h264-lab.h:5263-5278).  The chain is serial, but almost no macroblock moves the state: 64 macroblocks per pass, a ballot
 * finds the next one that moves it or mismatches, everything in front of it is settled at once.  traj (optional) receives the
 * state in front of every macroblock.  Returns the first mismatching macroblock or -1; s = state behind the frame.
 */
/* macroblocks [k0, k1) of one slice, continuing from state s / verdict first_bad (the finalizer walks a frame row by row as its rows
 * complete; batches of 64 inside the range) */
DEV void clusters_walk_range(const h264e_frame_task_t &T, const GLOBAL_AS h264e_mbrec_t *rec, mv32 s[2], int &first_bad, GLOBAL_AS mv32 *traj, int k0, int k1)
{
    const GLOBAL_AS mv32 *per_mb = (const GLOBAL_AS mv32 *)T.clusters_per_mb;
    const mv32 u_frame0 = T.clusters[0], u_frame1 = T.clusters[1];
    for (int base = k0; base < k1; base += 64)
    {
        int start = 0;
        /* the state in front of every macroblock of the batch, one per lane (wave.h V64) */
        V64 mine0 = v64_make([&](int) -> int { return s[0]; }), mine1 = v64_make([&](int) -> int { return s[1]; });
        for (;;)
        {
            const mv32 c0 = s[0], c1 = s[1];
            const int fb = first_bad;
            const uint64_t ev = wave_ballot([&](int l) -> int {
                const int k = base + l;
                if (l < start || k >= k1) return 0;
                const mv32 mv0 = rec[k].mv0;
                const int type = rec[k].type;
                int hit = 0;
                if (fb < 0 && rec[k].used_cand)
                {
                    const mv32 u0 = per_mb ? per_mb[2*k] : u_frame0, u1 = per_mb ? per_mb[2*k + 1] : u_frame1;
                    if (mvround(u0) != mvround(c0) || mvround(u1) != mvround(c1)) hit = 1;
                }
                if (type < 5)
                {
                    mv32 c[2] = { c0, c1 };
                    clusters_step(c, mv0);
                    if (c[0] != c0 || c[1] != c1) hit = 1;
                }
                return hit;
            });
            if (!ev) break;
            const int e = __builtin_ctzll(ev), ke = base + e;
            /* macroblock ke: mismatch check first (against the state in front of it), then its update */
            if (first_bad < 0 && rec[ke].used_cand)
            {
                const mv32 u0 = per_mb ? per_mb[2*ke] : u_frame0, u1 = per_mb ? per_mb[2*ke + 1] : u_frame1;
                if (mvround(u0) != mvround(s[0]) || mvround(u1) != mvround(s[1])) first_bad = ke;
            }
            if (rec[ke].type < 5) clusters_step(s, rec[ke].mv0);
            s[0] = (mv32)uni(s[0]); s[1] = (mv32)uni(s[1]); first_bad = uni(first_bad);
            start = e + 1;
            mine0 = v64_map(mine0, [&](int l, int v) -> int { return l >= start ? s[0] : v; });
            mine1 = v64_map(mine1, [&](int l, int v) -> int { return l >= start ? s[1] : v; });
            if (start >= 64) break;
        }
        if (traj)
        {
            v64_each(mine0, [&](int l, int v) { const int k = base + l; if (k < k1) traj[2*k] = v; });
            v64_each(mine1, [&](int l, int v) { const int k = base + l; if (k < k1) traj[2*k + 1] = v; });
        }
    }
}

/* the whole frame at once (the walk of a complete frame: tests, tools) */
DEV int device_clusters_walk(const h264e_geom_t &G, const h264e_frame_task_t &T, const GLOBAL_AS h264e_mbrec_t *rec, mv32 s[2], GLOBAL_AS mv32 *traj)
{
    const mv32 s0[2] = { s[0], s[1] };
    int first_bad = -1;
    for (int band = 0; band < T.nslices; band++)
    {
        s[0] = s0[0]; s[1] = s0[1];
        clusters_walk_range(T, rec, s, first_bad, traj, T.slice_row[band]*G.nmbx, T.slice_row[band + 1]*G.nmbx);
    }
    if (T.nslices > 1) { s[0] = s0[0]; s[1] = s0[1]; }      /* the parent's state never moves in the row-band build (h264-lab.h:6526) */
    wave_sync();
    return first_bad;
}

/*
 * The finalizer's walk, ROW BY ROW as the rows complete (JobWalk::rows is called from the splice below, for the rows whose counters say
 * complete, once the state in front of the frame is known): the verdict of a frame is known a walk of ONE row after its last row ended.
 * The walk goes on behind a mismatch, over every row, as the whole-frame walk does: the trajectory behind the mismatch is the speculation
 * for the next encode of those rows, and it is a good one (the vectors of macroblocks that consumed slightly wrong candidates are mostly
 * the right ones).  Stopping the launch as soon as the row with the first mismatch has ended, with the state at the stop as the
 * trajectory of everything behind it, was built and measured in round 4: it wins a little where a frame fails once (1080p 4 Mbit/s:
 * +2.5 %) and loses badly where the state keeps moving through a frame (8K 60 Mbit/s, two slices: 50 launches instead of 22, the
 * re-encodes advance two or three rows at a time) -- not kept.
 */
struct JobWalk
{
    mv32 s0[2], s[2];
    int first_bad, next_row;
    DEVM void begin(const mv32 st[2]) { s0[0] = s[0] = st[0]; s0[1] = s[1] = st[1]; first_bad = -1; next_row = 0; }
    /* walks the rows up to and including `upto` that have not been walked yet (all of them complete) */
    DEVM void rows(const h264e_geom_t &G, const h264e_frame_task_t &T, const GLOBAL_AS h264e_mbrec_t *rec, GLOBAL_AS mv32 *traj, int upto)
    {
        for (; next_row <= upto; next_row++)
        {
            const int r = next_row;
            for (int k = 0; k < T.nslices; k++) if (r == T.slice_row[k]) { s[0] = s0[0]; s[1] = s0[1]; }      /* every slice starts from the state in front of the frame */
            clusters_walk_range(T, rec, s, first_bad, traj, r*G.nmbx, (r + 1)*G.nmbx);
        }
    }
    /* behind the last row: the state to hand on */
    DEVM void end(const h264e_frame_task_t &T)
    {
        if (T.nslices > 1) { s[0] = s0[0]; s[1] = s0[1]; }      /* the parent's state never moves in the row-band build (h264-lab.h:6526) */
        wave_sync();
    }
};

struct SpliceOut { uint32_t start, nbytes; int overflow, all_skipped; };

/* The slices' RBSPs from the row bit buffers.  on_row(first row of the slice, row) is called before a row is touched: the caller waits
 * for the row there (and walks it); a non-zero return ends the splice with that code (nothing is committed). */
template <class ROW> DEV int splice_frame(const h264e_geom_t &G, const ChainG &C, const h264e_frame_task_t &T, SpliceOut &o, ROW on_row)
{
    SpliceState s;
    const uint32_t start = T.arena_reset ? 0u : ((*C.cursor + 15u) & ~15u);
    uint32_t off = 0;                                                           /* of the current slice, relative to start */
    int overflow = 0, all_skipped = 0, spl_overflow = 0;
    GLOBAL_AS h264e_frameout_t &F = C.fout[T.frame_slot];
    /* one RBSP per slice (row band); the slices of a frame lie behind each other, each starting 16-byte aligned */
    for (int k = 0; k < T.nslices; k++)
    {
        const int r0 = T.slice_row[k], r1 = T.slice_row[k + 1];
        const uint32_t room = start + off < C.arena_cap ? C.arena_cap - (start + off) : 0u;
        s.out = (GLOBAL_AS uint32_t *)(C.arena + start + off);
        s.wpos = 0; s.carry = 0; s.cbits = 0; s.overflow = 0;
        s.cap_words = room >> 2;
        splice_put(s, 8, (uint64_t)(uint32_t)T.hdr_nal);
        put_ue64(s, (uint32_t)(r0*G.nmbx));                                     /* first_mb_in_slice, h264-lab.h:4247 */
        splice_put(s, T.hdr_nbits, T.hdr_bits);
        int run = 0;
        for (int row = r0; row < r1; row++)
        {
            const int stop = on_row(r0, row);
            if (stop) return stop;
            const GLOBAL_AS h264e_rowmeta_t &M = C.rowmeta[row];
            overflow |= M.overflow;
            run += M.lead_skips;
            if (M.lead_skips < G.nmbx || M.nbits)
            {
                if (T.slice_type != 2) put_ue64(s, (uint32_t)run);
                wave_sync();
                splice_words(s, C.rowbits + (size_t)row*G.row_words, M.nbits);
                wave_sync();
                run = M.trail_skips;
            }
        }
        /* rc_frame_end's skip flag (h264-lab.h:6596): in the row-band build the parent's skip_run is never touched, so it is 0 there */
        if (T.nslices == 1) all_skipped = run == G.nmb;
        if (run) put_ue64(s, (uint32_t)run);                                    /* h264-lab.h:6451-6454 */
        splice_put(s, 1, 1);                                                    /* rbsp_stop_one_bit, h264-lab.h:3999 */
        const uint32_t nbytes = s.wpos*4 + (uint32_t)((s.cbits + 7) >> 3);
        if (s.cbits) { int pad = 32 - s.cbits; splice_put(s, pad, 0); }
        spl_overflow |= s.overflow;
        F.slice_nbytes[k] = nbytes;
        if (k + 1 < T.nslices) off += (nbytes + 15u) & ~15u; else off += nbytes;
        wave_sync();
    }
    o.start = start; o.nbytes = off; o.overflow = overflow | spl_overflow; o.all_skipped = all_skipped;
    return 0;
}

/* ... and the frame's result record, once the frame stands */
DEV void finalize_commit(const h264e_geom_t &G, const ChainG &C, const h264e_frame_task_t &T, const SpliceOut &o, GLOBAL_AS int *stepflags)
{
    GLOBAL_AS h264e_frameout_t &F = C.fout[T.frame_slot];
    /* mv_clusters speculation check of the frame-at-a-time path (its host walks exactly when this says "moved"): is the speculated state
     * a fixed point of every update of this frame?  Frames validated by the device walk do not need it. */
    const GLOBAL_AS h264e_mbrec_t *rec = C.mbrec + (size_t)T.frame_slot*G.nmb;
    int moved = 0;
    if (!T.clusters_per_mb && !T.walk_on_device)
    {
        for (int base = 0; base < G.nmb; base += 64)
        {
            uint64_t bad = wave_ballot([&](int l) -> int {
                int k = base + l;
                if (k >= G.nmb || rec[k].type >= 5) return 0;
                mv32 c[2] = { T.clusters[0], T.clusters[1] };
                clusters_step(c, rec[k].mv0);
                return c[0] != T.clusters[0] || c[1] != T.clusters[1];
            });
            if (bad) moved = 1;
        }
    }
    F.offset = o.start;
    F.nbytes = o.nbytes;
    F.nslices = T.nslices;
    F.all_skipped = o.all_skipped;
    F.clusters_moved = moved;
    F.overflow = o.overflow;
    F.far_reads = g_atomic_load(C.far_reads);
    if (wave_lane() == 0) g_atomic_store(C.far_reads, 0);
    *C.cursor = o.start + ((o.nbytes + 15u) & ~15u);
    stepflags[0] = moved;
    stepflags[1] = o.overflow;
    wave_sync();
}

/*
 * RBSP -> Annex-B NAL on the device: 4-byte start code + payload with emulation prevention (h264-lab.h:3926-4022 nal_count_esc /
 * nal_put_esc: a 0x03 goes in front of every byte <= 3 that follows two zero bytes; the zero count restarts behind it).
 * The escape automaton is sequential, but it can only fire where the RAW bytes read 00 00 0x -- once per ~4 MB of CAVLC data.
 * So: 256-byte blocks, one dword per lane; a ballot finds the blocks that contain such a triple (two bytes of look-back across
 * the block edge); all other blocks are copied by the whole wave (shifted by the escapes inserted so far), a block with a
 * candidate is run through the automaton byte by byte by one lane with the exact carried zero count.
 * dst: 16-byte aligned (host-mapped memory or HBM).  Returns the NAL size; sets overflow when it does not fit cap.
 */
DEV uint32_t nal_escape_copy(GLOBAL_AS uint8_t *dst, uint32_t cap, const GLOBAL_AS uint8_t *src, uint32_t n, int &overflow)
{
    uint32_t esc = 0;           /* escapes inserted so far (wave-uniform) */
    int cntz = 0;               /* the automaton's zero count in front of the current block (exact whenever it matters) */
    if (n + 4u > cap) { overflow = 1; return 0; }
    if (wave_lane() == 0) gstore32((gu8 *)dst, 0x01000000u);                /* 00 00 00 01 */
    for (uint32_t base = 0; base < n; base += 256)
    {
        const uint32_t len = n - base < 256u ? n - base : 256u;
        const uint64_t cand = wave_ballot([&](int l) -> int {
            const uint32_t o = base + 4u*(uint32_t)l;
            if (o >= n) return 0;
            /* six raw bytes: two of look-back (0xff in front of the payload) and this lane's dword (the arena is zero-padded to a
             * dword behind the payload: at worst a false candidate, which only selects the byte-wise path for this block) */
            const uint32_t cur = gload32((const gu8 *)(src + o)), prev = o ? gload32((const gu8 *)(src + o - 4)) : 0xffffffffu;
            const uint64_t w = ((uint64_t)cur << 16) | (uint64_t)(prev >> 16);
            int hit = 0;
            for (int k = 0; k < 4; k++)
                if (((w >> (8*k)) & 0xffffu) == 0 && ((w >> (8*(k + 2))) & 0xffu) <= 3u) hit = 1;
            return hit;
        });
        if (!cand)
        {
            /* no escape can fire inside this block: whole-wave copy, shifted by the escapes so far */
            if (4u + base + esc + len > cap) { overflow = 1; return 0; }
            WAVE_FOR(l)
            {
                const uint32_t o = 4u*(uint32_t)l;
                if (o < len)
                {
                    GLOBAL_AS uint8_t *d = dst + 4u + base + esc + o;
                    if (o + 4u <= len) gstore32((gu8 *)d, gload32((const gu8 *)(src + base + o)));
                    else for (uint32_t k = o; k < len; k++) dst[4u + base + esc + k] = src[base + k];
                }
            }
            /* zero count behind a block without escapes = its trailing raw zeros (at most 2, else it had a candidate) */
            {
                const uint32_t e = base + len;
                cntz = src[e - 1] ? 0 : (e >= 2 && src[e - 2] == 0) ? 2 : 1;
            }
        } else
        {
            if (4u + base + esc + len + len/2 + 2u > cap) { overflow = 1; return 0; }
            if (wave_lane() == 0)
            {
                uint32_t j = 4u + base + esc;
                for (uint32_t i = 0; i < len; i++)
                {
                    const uint8_t b = src[base + i];
                    if (cntz == 2 && b <= 3) { dst[j++] = 3; cntz = 0; esc++; }
                    cntz = b ? 0 : cntz + 1;
                    dst[j++] = b;
                }
            }
            esc = (uint32_t)uni((int)esc);       /* lane 0 ran the automaton: its counters are the wave's */
            cntz = uni(cntz);
        }
        wave_sync();
    }
    return 4u + n + esc;
}

/* finished frame -> every slice as a complete Annex-B NAL (start code + escaped payload) in the slot's device NAL arena, the NALs
 * behind each other at 16-byte aligned offsets; then NALs (when they fit the host-mapped mirror, which is sized for ordinary
 * frames: in_device = 0) and macroblock records to host-mapped memory, 16 bytes per lane per pass.  Returns the NAL sizes. */
DEV void export_frame(const h264e_geom_t &G, const ChainG &C, const h264e_frame_task_t &T, uint32_t *nal_bytes /* [H264E_MAX_SLICES], uniform */, uint32_t &total, int &overflow, int &in_device)
{
    const GLOBAL_AS h264e_frameout_t &F = C.fout[T.frame_slot];
    uint32_t soff = 0, doff = 0;
    overflow = 0;
    for (int k = 0; k < H264E_MAX_SLICES; k++)
    {
        nal_bytes[k] = 0;
        if (k < F.nslices)
        {
            const uint32_t n = F.slice_nbytes[k];
            const uint32_t room = doff < C.nal_cap ? C.nal_cap - doff : 0u;
            const uint32_t w = nal_escape_copy(C.nal_arena + doff, room, C.arena + F.offset + soff, n, overflow);
            nal_bytes[k] = w;
            soff += (n + 15u) & ~15u;
            doff += (w + 15u) & ~15u;
        }
    }
    total = doff;
    in_device = doff > T.host_rbsp_cap;
    if (!in_device)
    {
        drain_stores();                          /* the NAL arena was written by this wave just now */
        wave_sync();
        const uint32_t nw = (doff + 15u) >> 4;
        const GLOBAL_AS u32x4 *src = (const GLOBAL_AS u32x4 *)C.nal_arena;
        GLOBAL_AS u32x4 *dst = (GLOBAL_AS u32x4 *)T.host_rbsp;
        for (uint32_t base = 0; base < nw; base += 64)
        {
            WAVE_FOR(l) { if (base + (uint32_t)l < nw) dst[base + l] = src[base + l]; }
        }
    }
    const uint32_t nr = ((uint32_t)G.nmb*(uint32_t)sizeof(h264e_mbrec_t) + 15u) >> 4;
    const GLOBAL_AS u32x4 *rs = (const GLOBAL_AS u32x4 *)(C.mbrec + (size_t)T.frame_slot*G.nmb);
    GLOBAL_AS u32x4 *rd = (GLOBAL_AS u32x4 *)T.host_mbrec;
    for (uint32_t base = 0; base < nr; base += 64)
    {
        WAVE_FOR(l) { if (base + (uint32_t)l < nr) rd[base + l] = rs[base + l]; }
    }
    wave_sync();
}

#endif
