/*
 * h264o_enc.c -- macroblock pipeline, slice/frame/sequence drivers of the CPU oracle.
 * TEST INFRASTRUCTURE (see h264o_internal.h).  "H:n" = /root/reference/src/h264-lab.h:n,
 * "T:n" = /root/reference/src/minih264e_test.c:n.
 *
 * Layout differs from the reference on purpose (one explicit encoder object, explicit
 * left/top neighbour contexts, content-addressed prediction buffers instead of pointer
 * juggling); the decisions, their order and every tie-break follow the reference.
 */
#include <stdlib.h>
#include "h264o_internal.h"
#include "h264o_tables.h"

#define GUARD 16
#define MUL_LAMBDA(x, l) ((x)*(l) >> 4)                 /* H:3235 */
#define SLICE_P 0
#define SLICE_I 2

typedef struct { int x0, y0, x1, y1; } rect_t;

struct h264o_enc
{
    h264o_param_t par;
    int w, h, nmbx, nmby, nmb, cropping;
    int stride[3];
    uint8_t *mem, *ref[3], *dec[3];
    rect_t mv_limit, mv_qlimit;                         /* H:6322-6325 */

    int frame_num, next_idr_pic_id, pic_init_qp, frames_done;
    int slice_start_num, slice_start_row;               /* H:4185 slice.start_mb_num (row bands, H:6526-6534) */
    mv32 clusters[2];                                   /* H:766 */

    int slice_type, qp, prev_qp, speed, no_deblock;
    uint16_t qdat[2][42];

    /* neighbour contexts: *_top arrays hold the bottom edge of the row above, per macroblock column */
    mv32 *mv_top, mv_left[4], mv_tl[4];                 /* H:742, layout H:3649-3715 */
    uint8_t *nnz_top, nnz_left[8];                      /* 4 Y + 2 U + 2 V per MB, H:4386-4401 */
    int8_t *i4_top, i4_left[4];                         /* H:744 */
    uint8_t *pix_top, pix_left[32], pix_tl[3];          /* unfiltered recon: 16 Y + 8 U + 8 V, H:4693-4714 */
    uint8_t *df_qp; int8_t *df_type; uint8_t *df_nz_top;/* H:590-606 */
    uint32_t df_nzflag;
    mv32 df_mv[25];

    const uint8_t *in[3];
    int in_stride[3];

    bitw_t bw;
    uint8_t *rbsp; size_t rbsp_cap;
    uint8_t *out; size_t out_cap, out_pos;
    int skip_run;
    h264o_mbtrace_t *trace;

    /* frame-level rate control (H:686-699) */
    struct { int vbv_bits, qp_smooth, dqp_smooth, max_dqp, bit_budget, vbv_target_level; } rc;
    int desired_frame_bytes, qp_min, qp_max;
};

typedef struct
{
    int x, y, num, avail, type, cost, i16_mode, cropped;
    int8_t i4_mode[16];
    mv32 mv[16], mvd[16], mv_skip_pred;
    uint8_t inp[256], inp_c[128];                       /* chroma: U columns 0..7, V columns 8..15 */
    uint8_t pred[256], pred_c[128], i4rec[256];
    qblk_t qy[16], qu[4], qv[4];
    int16_t dcy[16], dcu[4], dcv[4], lev_dcy[16], lev_dcu[4], lev_dcv[4];
    unsigned nz_mask;
} mb_t;

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int iabs(int x) { return x < 0 ? -x : x; }
static inline int in_rect(mv32 v, const rect_t *r)
{
    return mvy(v) >= r->y0 && mvy(v) <= r->y1 && mvx(v) >= r->x0 && mvx(v) <= r->x1;
}
static inline mv32 clip_rect(mv32 v, const rect_t *r)
{
    return mvmk(imin(imax(mvx(v), r->x0), r->x1), imin(imax(mvy(v), r->y0), r->y1));
}
static inline mv32 mb_abs(const mb_t *m, mv32 v) { return mvadd(v, mvmk(m->x*64, m->y*64)); }
static inline int mv_cost(const h264o_enc_t *e, mv32 v, mv32 pred)       /* H:4952-4956 */
{
    return MUL_LAMBDA(bits_se(mvx(v) - mvx(pred)) + bits_se(mvy(v) - mvy(pred)), k_lambda_mv_q4[e->qp]);
}

/* ------------------------------------------------------------------ MV prediction */

/* H:3605-3622 */
static int avail_flags(const h264o_enc_t *e, const mb_t *m)
{
    const int s0 = e->slice_start_num;
    int f = m->num >= s0 + e->nmbx;
    if (m->num >= s0 + e->nmbx - 1 && m->x != e->nmbx - 1) f += AV_TR;
    if (m->num != s0 && m->x) f += AV_L;
    if (m->num > s0 + e->nmbx && m->x) f += AV_TL;
    return f;
}

static int med3(int a, int b, int c) { return imax(imin(imax(a, b), c), imin(a, b)); }

/* H:3696-3715 me_mv_medianpredictor_put (x,y,w,h in 4x4-block units) */
static void mvp_put(h264o_enc_t *e, const mb_t *m, int x, int y, int w, int h, mv32 mv)
{
    mv32 *top = e->mv_top + 4*m->x;
    int i;
    e->mv_tl[y] = top[x + w - 1];
    for (i = 1; i < h; i++) e->mv_tl[y + i] = mv;
    for (i = 0; i < h; i++) e->mv_left[y + i] = mv;
    for (i = 0; i < w; i++) top[x + i] = mv;
}

/* H:3720-3872 me_mv_medianpredictor_get (x,y,w,h in 4x4-block units) */
static mv32 mvp_get(const h264o_enc_t *e, const mb_t *m, int x, int y, int w, int h)
{
    const mv32 *top = e->mv_top + 4*m->x;
    int flag = m->avail, type = 1;                      /* 1 median, 2 left, 3 up, 4 up-right */
    mv32 a = e->mv_left[y], b = top[x], c = top[x + w], d = e->mv_tl[y], ret = 0;
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
    switch (type)
    {
    default:
        if (!(OK(b) || OK(c))) { if (OK(a)) ret = a; }
        else
        {
            if (!OK(a)) a = 0;
            if (!OK(b)) b = 0;
            if (!OK(c)) c = 0;
            ret = mvmk(med3(mvx(a), mvx(b), mvx(c)), med3(mvy(a), mvy(b), mvy(c)));
        }
        break;
    case 2: if (OK(a)) ret = a; break;
    case 3: if (OK(b)) ret = b; break;
    case 4: if (OK(c)) ret = c; break;
    }
#undef OK
    return ret;
}

/* ------------------------------------------------------------------ motion search */

/* H:5181-5193 me_mv_set_range */
static void set_range(mv32 *pnt, rect_t *range, const rect_t *limit, int mby_q)
{
    rect_t r = *limit;
    r.y0 = (int16_t)imax(r.y0, mby_q - 63*4);
    r.y1 = (int16_t)imin(r.y1, mby_q + 63*4);
    *pnt = clip_rect(*pnt, &r);
    {
        mv32 tl = clip_rect(mvadd(*pnt, mvmk(-32*4, -32*4)), &r), br = clip_rect(mvadd(*pnt, mvmk(32*4, 32*4)), &r);
        range->x0 = mvx(tl); range->y0 = mvy(tl); range->x1 = mvx(br); range->y1 = mvy(br);
    }
}

static int sad_at(const uint8_t *ref, int stride, mv32 v, const uint8_t *b, int w, int h)
{
    return sad_wh(ref + (mvy(v) >> 2)*stride + (mvx(v) >> 2), stride, b, 16, w, h);
}

static void copy_wh16(uint8_t *d, const uint8_t *s, int w, int h)
{
    int y;
    for (y = 0; y < h; y++) memcpy(d + 16*y, s + 16*y, (size_t)w);
}

/*
 * H:4973-5176 me_search_diamond.  ref = reference luma at the partition's own position, so *mv
 * (absolute for the MACROBLOCK, quarter-pel) is a displacement of that pointer.  The uint16 SAD
 * cache and its 0xffff "not evaluated" sentinel are part of the behaviour.
 * out receives the w x h prediction (stride 16) of the returned vector.
 */
static int diamond(const h264o_enc_t *e, const uint8_t *ref, const uint8_t *b, int stride, mv32 *mv, const rect_t *range,
                   mv32 mv_pred, int min_sad, int w, int h, uint8_t *out)
{
    static const int dxy[4][2] = { { 4, 0 }, { -4, 0 }, { 0, 4 }, { 0, -4 } };
    uint16_t cache[8];
    int dir, cloop, dir_prev, cost, i;
    mv32 v;
restart:
    dir = 0; cloop = 4; dir_prev = -1;
    for (i = 0; i < 8; i++) cache[i] = 0xffff;
    do
    {
        v = mvadd(*mv, mvmk(dxy[dir][0], dxy[dir][1]));
        if (in_rect(v, range) && cache[dir] == 0xffffu)
        {
            cost = sad_at(ref, stride, v, b, w, h) + mv_cost(e, v, mv_pred);
            cache[dir] = (uint16_t)cost;
            if (cost < min_sad)
            {
                int corner = ~0;
                if (dir_prev >= 0) corner = cache[4 + dir];
                for (i = 0; i < 4; i++) { cache[4 + i] = cache[i]; cache[i] = 0xffff; }
                if (dir_prev >= 0) cache[dir_prev ^ 1] = (uint16_t)corner;
                cache[dir ^ 1] = (uint16_t)min_sad;
                dir_prev = dir;
                dir--;
                cloop = 4 + 1;
                *mv = v;
                min_sad = cost;
            }
        }
        dir = (dir + 1) & 3;
    } while (--cloop);

    {   /* one diagonal probe towards the better vertical and the better horizontal neighbour */
        int pri = cache[3] >= cache[2] ? 2 : 3, sec = cache[1] >= cache[0] ? 0 : 1;
        v = mvadd(*mv, mvmk(dxy[pri][0] + dxy[sec][0], dxy[pri][1] + dxy[sec][1]));
        if (in_rect(v, range))
        {
            cost = sad_at(ref, stride, v, b, w, h) + mv_cost(e, v, mv_pred);
            if (cost < min_sad)
            {
                *mv = v;
                min_sad = cost;
                goto restart;
            }
        }
    }

    interp_luma(ref, stride, mvx(*mv), mvy(*mv), w, h, out);
    if (e->speed < 9 && in_rect(*mv, &e->mv_qlimit))
    {
        uint8_t p00[256], p02[256], p20[256], p22[256], t[256];
        mv32 vbest = *mv, pq, sq, vd;
        unsigned ms1 = cache[1], ms2 = cache[3];
        copy_wh16(p00, out, w, h);
        sq = mvmk(-1, 0); pq = mvmk(0, -1);
        if (cache[3] >= cache[2]) { pq = mvmk(0, 1); ms2 = cache[2]; }
        if (cache[1] >= cache[0]) { sq = mvmk(1, 0); ms1 = cache[0]; }
        if (ms2 > ms1) { mv32 s = sq; sq = pq; pq = s; }
        vd = mvadd(pq, sq);
        for (i = 0; i < 7; i++)
        {
            const uint8_t *cand;
            switch (i)
            {
            case 0: v = mvadd(*mv, mvadd(pq, pq)); interp_luma(ref, stride, mvx(v), mvy(v), w, h, p02); cand = p02; break;
            case 1: v = mvadd(*mv, pq); avg_wh(p00, p02, t, w, h); cand = t; break;
            case 2: v = mvadd(*mv, mvadd(sq, sq)); interp_luma(ref, stride, mvx(v), mvy(v), w, h, p20); cand = p20; break;
            case 3: v = mvadd(*mv, sq); avg_wh(p00, p20, t, w, h); cand = t; break;
            case 4: v = mvadd(*mv, vd); avg_wh(p02, p20, t, w, h); cand = t; break;
            case 5: v = mvadd(*mv, mvadd(vd, vd)); interp_luma(ref, stride, mvx(v), mvy(v), w, h, p22); cand = p22; break;
            default: v = mvadd(*mv, mvadd(pq, vd)); avg_wh(p22, p02, t, w, h); cand = t; break;
            }
            cost = sad_wh(cand, 16, b, 16, w, h) + mv_cost(e, v, mv_pred);
            if (cost < min_sad)
            {
                min_sad = cost;
                vbest = v;
                copy_wh16(out, cand, w, h);
            }
        }
        *mv = vbest;
    }
    return min_sad;
}

/* H:5224-5257 mb_inter_partition */
static void partition_hints(const int sad[4], int mode[4])
{
    int sum = sad[0] + sad[1] + sad[2] + sad[3];
    int slope = iabs((sad[0] - sad[2]) + (sad[1] - sad[3])) - iabs((sad[0] - sad[1]) + (sad[2] - sad[3]));
    int skew = iabs(sad[3] - sad[0]) - iabs(sad[2] - sad[1]);
    if (slope > (sum >> 4)) mode[1] = 1;
    if (slope < -(sum >> 4)) mode[2] = 1;
    if (iabs(skew) > (sum >> 4) && iabs(slope) <= (sum >> 4)) mode[3] = 1;
}

/* H:4915-4947 interpolate_chroma: per partition, from the reference chroma planes into pred_c */
static void predict_chroma_inter(const h264o_enc_t *e, mb_t *m)
{
    int c, w = (m->type & 2) ? 4 : 8, h = (m->type & 1) ? 4 : 8;
    if (m->type == -1) w = h = 8;
    for (c = 1; c < 3; c++)
    {
        int part = 0, x = 0, y = 0;
        for (;; part++)
        {
            mv32 v = mb_abs(m, m->mv[part]);
            interp_chroma(e->ref[c] + y*e->stride[c] + x, e->stride[c], mvx(v), mvy(v), w, h, m->pred_c + (c - 1)*8 + 16*y + x);
            x = (x + w) & 7;
            if (!x)
            {
                y = (y + h) & 7;
                if (!y) break;
            }
        }
    }
}

/*
 * Chroma half of the early-skip test, H:5322-5349.  For cropped edge macroblocks the reference
 * copies the padded 8x8 input into mb_pix_store (H:5333), which at that moment is the very buffer
 * holding the chroma prediction (ptest after the swap of H:5316): rows 0..3 of the prediction are
 * overwritten by the input copy before the SAD is taken.  Reproduced here by construction.
 */
static int skip_chroma_ok(const h264o_enc_t *e, const mb_t *m)
{
    int c, thr = k_skip_thr_inter[e->qp];
    uint8_t pc[128];
    memcpy(pc, m->pred_c, 128);
    for (c = 0; c < 2; c++)
    {
        int x, y, sad = 0;
        if (m->cropped)
            for (y = 0; y < 8; y++) for (x = 0; x < 8; x++) pc[8*y + x] = m->inp_c[16*y + 8*c + x];
        for (y = 0; y < 8; y++) for (x = 0; x < 8; x++) sad += iabs(m->inp_c[16*y + 8*c + x] - pc[16*y + 8*c + x]);
        if (sad >= thr) return 0;
    }
    return 1;
}

/* H:5283-5524 inter_choose_mode */
static void inter_choose(h264o_enc_t *e, mb_t *m)
{
    static const int nbits[4] = { 1, 4, 4, 12 };
    int prefer[4] = { 1, 0, 0, 0 };
    mv32 cand[20], mv_skip, mv_skip_a, mv_pred16, mv_best = MV_NA;
    int sad, sad_skip = 0x7FFFFFFF, sad_best = 0x7FFFFFFF, cand_cost_best = 0, i, j = 0, ncand = 0, sad4[4];
    const uint8_t *ry = e->ref[0] + m->y*16*e->stride[0] + m->x*16;   /* reference luma at this macroblock */
    int rs = e->stride[0];
    const mv32 *top = e->mv_top + 4*m->x;
    uint8_t skip_pred[256], test[256];
    mv32 ctx_left[4], ctx_tl[4], ctx_top[4], part_mv[4][16], part_mvd[4][16];
    int t;

    /* H:3877-3890 skip predictor */
    mv_pred16 = mvp_get(e, m, 0, 0, 4, 4);
    m->mv_skip_pred = 0;
    if (!(~m->avail & (AV_L | AV_T)) && e->mv_left[0] != 0 && top[0] != 0) m->mv_skip_pred = mv_pred16;
    mv_skip = m->mv_skip_pred;
    mv_skip_a = mb_abs(m, mv_skip);

    for (i = 0; i < 4; i++)
    {
        e->df_mv[4 + 5*i] = e->mv_left[i];
        e->df_mv[i] = top[i];
    }

    if (in_rect(mv_skip_a, &e->mv_qlimit))
    {
        interp_luma(ry, rs, mvx(mv_skip), mvy(mv_skip), 16, 16, skip_pred);
        sad_skip = sad_16x16_q(m->inp, 16, skip_pred, 16, sad4);
        if (imax(imax(sad4[0], sad4[1]), imax(sad4[2], sad4[3])) < k_skip_thr_inter[e->qp])
        {
            m->type = -1;
            m->mv[0] = mv_skip;
            m->cost = 0;
            predict_chroma_inter(e, m);
            if (skip_chroma_ok(e, m))
            {
                memcpy(m->pred, skip_pred, 256);
                return;
            }
        }
        if (e->speed < 1) partition_hints(sad4, prefer);
        mv_best = cand[ncand++] = mvround(mv_skip);
        if (!((mvx(mv_skip) | mvy(mv_skip)) & 3))
        {
            sad_best = sad_skip;
            cand_cost_best = mv_cost(e, mv_skip, mv_pred16);
            j = 1;
        }
    }

    cand[ncand++] = mv_pred16;
    cand[ncand++] = 0;                                                /* H:3895-3914 */
    if ((m->avail & AV_L) && e->mv_left[0] != MV_NA) cand[ncand++] = e->mv_left[0];
    if ((m->avail & AV_T) && top[0] != MV_NA) cand[ncand++] = top[0];
    if ((m->avail & AV_TR) && top[4] != MV_NA) cand[ncand++] = top[4];
    if (m->x <= 0) cand[ncand++] = mvmk(8*4, 0);
    if (m->y <= 0) cand[ncand++] = mvmk(0, 8*4);
    cand[ncand++] = e->clusters[0];
    cand[ncand++] = e->clusters[1];

    {   /* H:5198-5218 round to full-pel, drop duplicates */
        int k = 1, n;
        cand[0] = mvround(cand[0]);
        for (n = 1; n < ncand; n++)
        {
            mv32 v = mvround(cand[n]);
            for (i = 0; i < k; i++) if (cand[i] == v) break;
            if (i == k) cand[k++] = v;
        }
        ncand = k;
    }

    for (; j < ncand; j++)
    {
        mv32 va = mb_abs(m, cand[j]);
        if (in_rect(va, &e->mv_limit))
        {
            int c = mv_cost(e, cand[j], mv_pred16), s4[4];
            sad = sad_16x16_q(ry + (mvy(cand[j]) >> 2)*rs + (mvx(cand[j]) >> 2), rs, m->inp, 16, s4);
            if (e->speed < 1) partition_hints(s4, prefer);
            if (sad + c < sad_best + cand_cost_best)
            {
                cand_cost_best = c;
                sad_best = sad;
                mv_best = cand[j];
            }
        }
    }
    sad_best += mv_cost(e, mv_best, mv_pred16);

    /* H:3646-3671 save the predictor context; every partitioning is tried from the same state */
    for (i = 0; i < 4; i++) { ctx_left[i] = e->mv_left[i]; ctx_tl[i] = e->mv_tl[i]; ctx_top[i] = top[i]; }
    m->cost = 0xffffff;
    for (t = 0; t < 4; t++)
    {
        int imv = 0, part_sad = MUL_LAMBDA(nbits[t], k_lambda_q4[e->qp]);
        int w = (t & 2) ? 8 : 16, h = (t & 1) ? 8 : 16, px = 0, py = 0;
        if (!prefer[t]) continue;
        for (;;)
        {
            rect_t range;
            uint8_t blk[256];
            mv32 mv, mvp, mvabs = mb_abs(m, mv_best);
            const uint8_t *pref = e->ref[0] + py*rs + px;           /* plane origin + partition offset: mvabs is absolute */
            set_range(&mvabs, &range, &e->mv_limit, m->y*64 + py*4);
            mvp = mvp_get(e, m, px >> 2, py >> 2, w >> 2, h >> 2);
            if (t)
            {
                mvabs = mvround(mb_abs(m, mvp));
                set_range(&mvabs, &range, &e->mv_limit, m->y*64 + py*4);
                sad_best = sad_at(pref, rs, mvabs, m->inp + py*16 + px, w, h) + mv_cost(e, mvabs, mb_abs(m, mvp));
            }
            part_sad += diamond(e, pref, m->inp + py*16 + px, rs, &mvabs, &range, mb_abs(m, mvp), sad_best, w, h, blk);
            {
                int y;
                for (y = 0; y < h; y++) memcpy(test + (py + y)*16 + px, blk + 16*y, (size_t)w);
            }
            mv = mvsub(mvabs, mvmk(m->x*64, m->y*64));
            part_mvd[t][imv] = mvsub(mv, mvp);
            part_mv[t][imv++] = mv;
            mvp_put(e, m, px >> 2, py >> 2, w >> 2, h >> 2, mv);
            px = (px + w) & 15;
            if (!px)
            {
                py = (py + h) & 15;
                if (!py) break;
            }
        }
        for (i = 0; i < 4; i++) { e->mv_left[i] = ctx_left[i]; e->mv_tl[i] = ctx_tl[i]; e->mv_top[4*m->x + i] = ctx_top[i]; }
        if (part_sad < m->cost)
        {
            memcpy(m->pred, test, 256);
            m->cost = part_sad;
            m->type = t;
            memcpy(m->mv, part_mv[t], (size_t)imv*sizeof(mv32));
            memcpy(m->mvd, part_mvd[t], (size_t)imv*sizeof(mv32));
        }
    }

    if (m->cost > sad_skip)
    {
        m->type = 0;
        m->cost = sad_skip + mv_cost(e, mv_skip, mv_pred16);
        m->mv[0] = mv_skip;
        m->mvd[0] = mvsub(mv_skip, mv_pred16);
        memcpy(m->pred, skip_pred, 256);
    }
}

/* H:5263-5278 mv_clusters_update */
static void clusters_update(h264o_enc_t *e, mv32 mv)
{
    int n = mvx(mv)*mvx(mv) + mvy(mv)*mvy(mv);
    int n0 = mvx(e->clusters[0])*mvx(e->clusters[0]) + mvy(e->clusters[0])*mvy(e->clusters[0]);
    int n1 = mvx(e->clusters[1])*mvx(e->clusters[1]) + mvy(e->clusters[1])*mvy(e->clusters[1]);
#define SMOOTH(c) c = mvmk((63*mvx(c) + mvx(mv) + 32) >> 6, (63*mvy(c) + mvy(mv) + 32) >> 6)
    if (n < n1) SMOOTH(e->clusters[0]);
    if (n >= n0) SMOOTH(e->clusters[1]);
#undef SMOOTH
}

/* ------------------------------------------------------------------ intra decisions */

/* H:4838-4858 intra_estimate_16x16 + H:4876-4896 intra_choose_16x16 */
static void intra16_choose(h264o_enc_t *e, mb_t *m, const uint8_t *left, const uint8_t *top)
{
    static const uint8_t valid[8] = { 4, 5, 6, 7, 4, 5, 6, 15 };
    const uint8_t *p = m->inp;
    int v = valid[m->avail & 7], mode, sad, sad4[4];
    int dx = iabs(p[0] - p[15]) + iabs(p[15*16] - p[15*16 + 15]) + iabs(p[8*16] - p[8*16 + 15]);
    int dy = iabs(p[0] - p[15*16]) + iabs(p[15] - p[15*16 + 15]) + iabs(p[8] - p[15*16 + 8]);
    uint8_t pr[256];
    if (dx > 30 + 3*dy && dy < (100 + 50 - e->qp) && (v & 1)) mode = 0;
    else if (dy > 30 + 3*dx && dx < (100 + 50 - e->qp) && (v & 2)) mode = 1;
    else mode = 2;
    m->i16_mode = mode;
    pred16(pr, left, top, mode);
    sad = sad_16x16_q(m->inp, 16, pr, 16, sad4) + MUL_LAMBDA(bits_ue(mode + 1), k_lambda_q4[e->qp]) + k_lambda_i16_q4[e->qp];
    if (sad < m->cost)
    {
        m->cost = sad;
        m->type = 6;
        memcpy(m->pred, pr, 256);
    }
}

/* H:4723-4833 intra_choose_4x4: 16 blocks in raster order, each predicted from reconstructed neighbours */
static void intra4_choose(h264o_enc_t *e, mb_t *m)
{
    static const uint8_t block2avail[16] = { 0x07, 0x23, 0x23, 0x2b, 0x9b, 0x77, 0xff, 0x77, 0x9b, 0xff, 0xff, 0x77, 0x9b, 0x77, 0xff, 0x77 };
    /* rec: 17 rows x 24 columns working picture, origin at (1,1): row 0 = top line incl. top-left and 4 top-right samples */
    uint8_t rec[17*24], *r0 = rec + 24 + 1;
    const uint8_t *top = e->pix_top + 32*m->x;
    int n, i, cost = k_lambda_i4_q4[e->qp], avail = m->avail;
    unsigned nz_mask = 0;
    int8_t *ctx_t = e->i4_top + 4*m->x;
    memset(rec, 0, sizeof(rec));
    r0[-24 - 1] = e->pix_tl[0];
    for (i = 0; i < 16; i++) r0[-24 + i] = top[i];
    for (i = 0; i < 4; i++) r0[-24 + 16 + i] = top[32 + i];          /* first 4 luma samples of the next macroblock's top line */
    for (i = 0; i < 16; i++) r0[24*i - 1] = e->pix_left[i];

    for (n = 0; n < 16; n++)
    {
        int r = n >> 2, c = n & 3, a, mpred, mode, sad, coded, x, y;
        uint8_t *blk = r0 + 24*4*r + 4*c, left4[4], pr[64];
        const uint8_t *bin = m->inp + (c + r*16)*4;
        a = (avail & block2avail[n]) | (block2avail[n] >> 4);
        if (!(block2avail[n] & AV_TL))
            if ((n <= 3 && (avail & AV_T)) || (n > 3 && (avail & AV_L))) a |= AV_TL;
        if (n < 3 && (avail & AV_T)) a |= AV_TR;

        mpred = imin(e->i4_left[r], ctx_t[c]);
        if (mpred < 0) mpred = 2;
        for (i = 0; i < 4; i++) left4[i] = blk[24*i - 1];
        mode = i4_choose(bin, pr, a, blk - 24, left4, blk[-24 - 1], mpred, MUL_LAMBDA(3, k_lambda_q4[e->qp]), &sad);
        e->i4_left[r] = ctx_t[c] = (int8_t)mode;
        m->i4_mode[n] = (int8_t)(mode == mpred ? -1 : mode > mpred ? mode - 1 : mode);

        coded = 0;
        if (sad > k_skip_thr_i4x4[e->qp])
        {
            coded = xform_quant(bin, 16, pr, QMODE_I4, m->qy + n, NULL, e->qdat[0]);
            if (coded) recon_blocks(pr, 16, pr, m->qy + n, 1, 0x80000000u);
        } else
            memset(m->qy + n, 0, sizeof(m->qy[0]));
        nz_mask = (nz_mask << 1) | (unsigned)coded;
        cost += sad;
        for (y = 0; y < 4; y++) for (x = 0; x < 4; x++) blk[24*y + x] = pr[16*y + x];
    }
    m->nz_mask = nz_mask & 0xffff;
    for (i = 0; i < 16; i++) memcpy(m->i4rec + 16*i, r0 + 24*i, 16);
    if (cost < m->cost)
    {
        m->cost = cost;
        m->type = 5;
    }
}

/* ------------------------------------------------------------------ macroblock write */

/* H:4378-4715 mb_write */
static void mb_write(h264o_enc_t *e, mb_t *m)
{
    static const uint8_t scan8[16] = { 0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15 };    /* H:920 */
    static const uint8_t cbp2code[2][48] = {                                                        /* Table 9-4 */
        { 3, 29, 30, 17, 31, 18, 37, 8, 32, 38, 19, 9, 20, 10, 11, 2, 16, 33, 34, 21, 35, 22, 39, 4,
          36, 40, 23, 5, 24, 6, 7, 1, 41, 42, 43, 25, 44, 26, 46, 12, 45, 47, 27, 13, 28, 14, 15, 0 },
        { 0, 2, 3, 7, 4, 8, 17, 13, 5, 18, 9, 14, 10, 15, 16, 11, 1, 32, 33, 36, 34, 37, 44, 40,
          35, 45, 38, 41, 39, 42, 43, 19, 6, 24, 25, 20, 26, 21, 46, 28, 27, 47, 22, 29, 23, 30, 31, 12 } };
    uint8_t *nnz_top = e->nnz_top + 8*m->x, *nnz_left = e->nnz_left, nz[9];
    uint8_t *dy = e->dec[0] + (m->y*16)*e->stride[0] + m->x*16;
    uint8_t *dc[2];
    int i, c, i16 = m->type >= 6, cbpl = 0, cbpc = 0, cbp, mb_type;
    bitw_t *b = &e->bw;
    dc[0] = e->dec[1] + (m->y*8)*e->stride[1] + m->x*8;
    dc[1] = e->dec[2] + (m->y*8)*e->stride[2] + m->x*8;

    if (m->type != 5) memset(e->i4_left, 2, 4), memset(e->i4_top + 4*m->x, 2, 4);

    e->df_nzflag = ((e->df_nzflag >> 4) & 0x84210) | e->df_nz_top[m->x];
    for (i = 0; i < 4; i++)
    {
        nz[5 + i] = nnz_top[i]; nnz_top[i] = 0;
        nz[3 - i] = nnz_left[i]; nnz_left[i] = 0;
    }
    nz[4] = 0;

    if (m->type != -1)
    {
        if (m->type != 5)
        {
            unsigned mask = (unsigned)xform_quant(m->inp, 16, m->pred, i16 ? QMODE_I16 : QMODE_INTER, m->qy, m->dcy, e->qdat[0]);
            m->nz_mask = mask & 0xffff;
            if (i16)
            {
                quant_luma_dc(m->qy, m->dcy, m->lev_dcy, e->qdat[0]);
                mask = 0xFFFF;
            }
            recon_blocks(dy, e->stride[0], m->pred, m->qy, 4, mask << 16);
        } else
        {
            for (i = 0; i < 16; i++) memcpy(dy + i*e->stride[0], m->i4rec + 16*i, 16);
        }
        if (m->nz_mask & 0xCC00) cbpl |= 1;
        if (m->nz_mask & 0x3300) cbpl |= 2;
        if (m->nz_mask & 0x00CC) cbpl |= 4;
        if (m->nz_mask & 0x0033) cbpl |= 8;

        for (c = 0; c < 2; c++)
        {
            qblk_t *q = c ? m->qv : m->qu;
            const uint8_t *pred = m->pred_c + 8*c;
            unsigned mask = (unsigned)xform_quant(m->inp_c + 8*c, 16, pred, QMODE_CHROMA, q, c ? m->dcv : m->dcu, e->qdat[1]);
            int dc_flag;
            if (mask) cbpc = 2;
            cbpc |= dc_flag = quant_chroma_dc(q, c ? m->dcv : m->dcu, c ? m->lev_dcv : m->lev_dcu, e->qdat[1]);
            if (dc_flag)
            {
                for (i = 0; i < 4; i++)
                    if (~mask & (8u >> i)) memset(q[i].dq + 1, 0, 15*sizeof(int16_t));
                mask = 15;
            }
            recon_blocks(dc[c], e->stride[1 + c], pred, q, 2, mask << 28);
        }
        cbpc = imin(cbpc, 2);

        /* roll back to skip: P16x16, nothing coded, vector equals the skip predictor (H:4493-4499) */
        if (!(m->type | cbpl | cbpc) && m->mv[0] == m->mv_skip_pred) m->type = -1;
    }

    if (m->type == -1)
    {
        int y;
        e->skip_run++;
        memset(nnz_top + 4, 0, 4); memset(nnz_left + 4, 0, 4);
        mvp_put(e, m, 0, 0, 4, 4, m->mv[0]);
        for (i = 0; i < 16; i++) e->df_mv[5 + 5*(i >> 2) + (i & 3)] = m->mv[0];
        for (y = 0; y < 16; y++) memcpy(dy + y*e->stride[0], m->pred + 16*y, 16);
        for (c = 0; c < 2; c++)
            for (y = 0; y < 8; y++) memcpy(dc[c] + y*e->stride[1 + c], m->pred_c + 16*y + 8*c, 8);
        cbp = 0;
    } else
    {
        mb_type = m->type;
        if (i16)
        {
            if (cbpl) cbpl = 15;
            mb_type += m->i16_mode + cbpc*4 + (cbpl ? 12 : 0);
        }
        if (mb_type >= 5 && e->slice_type == SLICE_I) mb_type -= 5;
        if (e->slice_type != SLICE_I)
        {
            bw_ue(b, (uint32_t)e->skip_run);
            e->skip_run = 0;
        }
        bw_ue(b, (uint32_t)mb_type);
        if (m->type == 3) for (i = 0; i < 4; i++) bw_ue(b, 0);
        if (m->type >= 5)
        {
            int cm;
            if (m->type == 5)
                for (i = 0; i < 16; i++)
                {
                    int md = m->i4_mode[scan8[i]];
                    if (md < 0) bw_put(b, 1, 1); else bw_put(b, 4, (uint32_t)md);
                }
            cm = m->i16_mode;
            if (!(cm & 1)) cm ^= 2;
            bw_ue(b, (uint32_t)cm);
            mvp_put(e, m, 0, 0, 4, 4, MV_NA);
        } else
        {
            int part, x = 0, y = 0, dx = (m->type & 2) ? 2 : 4, dyb = (m->type & 1) ? 2 : 4, xx, yy;
            for (part = 0;; part++)
            {
                bw_se(b, mvx(m->mvd[part]));
                bw_se(b, mvy(m->mvd[part]));
                mvp_put(e, m, x, y, dx, dyb, m->mv[part]);
                for (yy = 0; yy < dyb; yy++) for (xx = 0; xx < dx; xx++) e->df_mv[5 + 5*(y + yy) + x + xx] = m->mv[part];
                x = (x + dx) & 3;
                if (!x)
                {
                    y = (y + dyb) & 3;
                    if (!y) break;
                }
            }
        }
        cbp = cbpl + (cbpc << 4);
        if (!i16) bw_ue(b, cbp2code[m->type < 5][cbp]);
        if (cbp || i16)
        {
            bw_se(b, e->qp - e->prev_qp);
            e->prev_qp = e->qp;
        }
        if (i16) cavlc_block(b, m->lev_dcy, 0, 16, nz[3] + nz[5], &nz[4]);
        if (cbpl)
        {
            for (i = 0; i < 16; i++)
            {
                int j = scan8[i];
                uint8_t *pnz = nz + 4 + (j & 3) - (j >> 2);
                if (cbp & (1 << (i >> 2)))
                {
                    cavlc_block(b, m->qy[j].qv, i16, 16 - i16, pnz[-1] + pnz[1], pnz);
                    if (*pnz) e->df_nzflag |= 1u << (5 + (j & 3) + 5*(j >> 2));
                } else
                    *pnz = 0;
            }
            for (i = 0; i < 4; i++)
            {
                nnz_top[i] = nz[1 + i];
                nnz_left[i] = nz[7 - i];
            }
        }
        if (cbpc)
        {
            uint8_t dummy;
            cavlc_block(b, m->lev_dcu, 0, 4, 17 + 17, &dummy);
            cavlc_block(b, m->lev_dcv, 0, 4, 17 + 17, &dummy);
            if (cbpc > 1)
                for (c = 0; c < 2; c++)
                {
                    uint8_t nzc[5];
                    int off = c ? 6 : 4;
                    qblk_t *q = c ? m->qv : m->qu;
                    nzc[2] = 0;
                    for (i = 0; i < 2; i++)
                    {
                        nzc[3 + i] = nnz_top[off + i];
                        nzc[1 - i] = nnz_left[off + i];
                    }
                    for (i = 0; i < 4; i++)
                    {
                        int k = 2 + (i & 1) - (i >> 1);
                        cavlc_block(b, q[i].qv, 1, 15, nzc[k - 1] + nzc[k + 1], nzc + k);
                    }
                    for (i = 0; i < 2; i++)
                    {
                        nnz_top[off + i] = nzc[1 + i];
                        nnz_left[off + i] = nzc[3 - i];
                    }
                }
        }
        if (cbpc != 2) { memset(nnz_top + 4, 0, 4); memset(nnz_left + 4, 0, 4); }
    }

    {   /* H:4693-4714 keep the UNFILTERED right column / bottom row for intra prediction of later macroblocks */
        uint8_t *top = e->pix_top + 32*m->x;
        for (c = 0; c < 3; c++)
        {
            int n = c ? 8 : 16, off = c ? 8 + 8*c : 0;
            const uint8_t *p = c ? dc[c - 1] : dy;
            int s = e->stride[c];
            e->pix_tl[c] = top[off + n - 1];
            for (i = 0; i < n; i++)
            {
                e->pix_left[off + i] = p[n - 1 + i*s];
                top[off + i] = p[(n - 1)*s + i];
            }
        }
    }
    if (e->trace)
    {
        h264o_mbtrace_t *t = e->trace + m->num;
        t->type = (int8_t)m->type;
        t->cbp = (uint8_t)cbp;
        t->mvx = m->type < 5 ? (int16_t)mvx(m->mv[0]) : 0;
        t->mvy = m->type < 5 ? (int16_t)mvy(m->mv[0]) : 0;
        t->bitpos = (uint32_t)bw_bits(b);
    }
}

/* ------------------------------------------------------------------ deblock control */

/* H:5535-5637 df_strength + the edge masking of H:5653-5661 */
static void mb_deblock(h264o_enc_t *e, const mb_t *m)
{
    uint8_t bs[32];
    uint32_t flag = e->df_nzflag;
    int x, y, qp_top, qp_left, qp = e->prev_qp;
    memset(bs, 0, sizeof(bs));
    e->df_nz_top[m->x] = (uint8_t)(flag >> 20);
    if (m->type < 5)
    {
        const mv32 *mv = e->df_mv;
        for (y = 0; y < 4; y++, flag >>= 1, mv++)
            for (x = 0; x < 4; x++, flag >>= 1, mv++)
            {
                bs[4*x + y] = (flag & (3 << 4)) ? 2 :
                    (iabs(mvx(mv[4]) - mvx(mv[5])) > 3 || iabs(mvy(mv[4]) - mvy(mv[5])) > 3) ? 1 : 0;
                bs[16 + 4*y + x] = (flag & 33) ? 2 :
                    (iabs(mvx(mv[0]) - mvx(mv[5])) > 3 || iabs(mvy(mv[0]) - mvy(mv[5])) > 3) ? 1 : 0;
            }
    } else
    {
        memset(bs + 4, 3, 12);
        memset(bs + 20, 3, 12);
    }
    if (m->type >= 5 || (m->x && e->df_type[m->x - 1] >= 5)) memset(bs, 4, 4);
    if (m->type >= 5 || e->df_type[m->x] >= 5) memset(bs + 16, 4, 4);
    e->df_type[m->x] = (int8_t)m->type;
    if (!m->x) memset(bs, 0, 4);
    if (m->y == e->slice_start_row) memset(bs + 16, 0, 4);          /* picture top, or H:5799-5803: no filtering across a slice border */
    qp_top = e->df_qp[m->x];
    qp_left = m->x ? e->df_qp[m->x - 1] : qp;
    e->df_qp[m->x] = (uint8_t)qp;
    deblock_mb(e->dec[0] + m->y*16*e->stride[0] + m->x*16, e->stride[0],
               e->dec[1] + m->y*8*e->stride[1] + m->x*8, e->dec[2] + m->y*8*e->stride[2] + m->x*8, e->stride[1],
               bs, qp, qp_left, qp_top);
}

/* ------------------------------------------------------------------ macroblock driver */

/* H:3536-3562 pix_copy_cropped_mb semantics: replicate the last valid column / row */
static void load_block(uint8_t *d, int ds, int n, const uint8_t *s, int ss, int vw, int vh)
{
    int x, y;
    for (y = 0; y < n; y++)
        for (x = 0; x < n; x++)
            d[y*ds + x] = s[imin(y, vh - 1)*ss + imin(x, vw - 1)];
}

/* H:5724-5812 mb_encode */
static void encode_mb(h264o_enc_t *e, int mbx, int mby)
{
    mb_t mb, *m = &mb;
    const uint8_t *left = e->pix_left, *top = e->pix_top + 32*mbx;
    int vw, vh;
    memset(m, 0, sizeof(*m));
    m->x = mbx; m->y = mby; m->num = mby*e->nmbx + mbx;
    m->avail = avail_flags(e, m);
    m->cropped = e->cropping && ((mbx + 1)*16 > e->par.width || (mby + 1)*16 > e->par.height);
    vw = imin(16, e->par.width - mbx*16); vh = imin(16, e->par.height - mby*16);
    load_block(m->inp, 16, 16, e->in[0] + mby*16*e->in_stride[0] + mbx*16, e->in_stride[0], vw, vh);
    vw = imin(8, e->par.width/2 - mbx*8); vh = imin(8, e->par.height/2 - mby*8);
    load_block(m->inp_c, 16, 8, e->in[1] + mby*8*e->in_stride[1] + mbx*8, e->in_stride[1], vw, vh);
    load_block(m->inp_c + 8, 16, 8, e->in[2] + mby*8*e->in_stride[2] + mbx*8, e->in_stride[2], vw, vh);
    if (!(m->avail & AV_L)) left = NULL;
    if (!(m->avail & AV_T)) top = NULL;
    m->type = 0;
    m->cost = 0x7FFFFFFF;

    if (e->slice_type == SLICE_P) inter_choose(e, m);
    if (m->type >= 0)
    {
        intra16_choose(e, m, left, top);
        if (e->speed < 2 || e->slice_type != SLICE_P) intra4_choose(e, m);
    }
    if (m->type < 5) clusters_update(e, m->mv[0]);
    if (m->type >= 5) pred_chroma(m->pred_c, left ? left + 16 : NULL, top ? top + 16 : NULL, m->i16_mode);
    else predict_chroma_inter(e, m);
    mb_write(e, m);
    if (!e->no_deblock) mb_deblock(e, m);
}

/* ------------------------------------------------------------------ NAL / headers */

/* H:3926-4022: start code + payload with emulation prevention (0x03 after two zero bytes before 00..03) */
static void nal_emit(h264o_enc_t *e, const uint8_t *p, size_t n)
{
    size_t i;
    int zeros = 0;
    uint8_t *d = e->out + e->out_pos;
    d[0] = d[1] = d[2] = 0; d[3] = 1; d += 4;
    for (i = 0; i < n; i++)
    {
        if (zeros == 2 && p[i] <= 3) { *d++ = 3; zeros = 0; }
        zeros = p[i] ? 0 : zeros + 1;
        *d++ = p[i];
    }
    e->out_pos = (size_t)(d - e->out);
}

static void nal_begin(h264o_enc_t *e, int hdr) { bw_init(&e->bw, e->rbsp, e->rbsp_cap); bw_put(&e->bw, 8, (uint32_t)hdr); }
static void nal_finish(h264o_enc_t *e)
{
    bw_put(&e->bw, 1, 1);
    bw_flush(&e->bw);
    nal_emit(e, e->rbsp, e->bw.pos);
}

/* H:4040-4141 encode_sps (baseline, profile 66) */
static void write_sps(h264o_enc_t *e)
{
    static const struct { uint8_t level; uint16_t max_fs, max_vbvdiv5; uint32_t max_dpb; } lim[] = {
        { 10, 99, 175/5, 396 }, { 10, 99, 350/5, 396 }, { 11, 396, 500/5, 900 }, { 12, 396, 1000/5, 2376 },
        { 13, 396, 2000/5, 2376 }, { 20, 396, 2000/5, 2376 }, { 21, 792, 4000/5, 4752 }, { 22, 1620, 4000/5, 8100 },
        { 30, 1620, 10000/5, 8100 }, { 31, 3600, 14000/5, 18000 }, { 32, 5120, 20000/5, 20480 }, { 40, 8192, 25000/5, 32768 },
        { 41, 8192, 62500/5, 32768 }, { 42, 8704, 62500/5, 34816 }, { 50, 22080, 135000/5, 110400 }, { 51, 36864, 240000/5, 184320 } };
    int k = 0;
    bitw_t *b = &e->bw;
    while (lim[k].level < 51 && (e->nmb > lim[k].max_fs || e->par.vbv_size_bytes > lim[k].max_vbvdiv5*(5*1000/8) ||
                                 (unsigned)e->nmb > lim[k].max_dpb)) k++;
    nal_begin(e, 0x67);
    bw_put(b, 8, 66);
    bw_put(b, 8, 0);                        /* constraint flags: 0xE0/0xF0 & 4 == 0 */
    bw_put(b, 8, lim[k].level);
    bw_ue(b, 0);                            /* sps_id */
    bw_ue(b, 1);                            /* log2_max_frame_num_minus4 */
    bw_ue(b, 2);                            /* pic_order_cnt_type */
    bw_ue(b, 1);                            /* num_ref_frames */
    bw_put(b, 1, 0);
    bw_ue(b, (uint32_t)(((e->par.width + 15) >> 4) - 1));
    bw_ue(b, (uint32_t)(((e->par.height + 15) >> 4) - 1));
    bw_put(b, 3, (uint32_t)(6 + e->cropping));
    if (e->cropping)
    {
        bw_ue(b, 0); bw_ue(b, (uint32_t)((e->w - e->par.width) >> 1));
        bw_ue(b, 0); bw_ue(b, (uint32_t)((e->h - e->par.height) >> 1));
    }
    bw_put(b, 1, 0);
    nal_finish(e);
}

/* H:4147-4176 encode_pps */
static void write_pps(h264o_enc_t *e)
{
    bitw_t *b = &e->bw;
    nal_begin(e, 0x68);
    bw_ue(b, 0); bw_ue(b, 0);
    bw_put(b, 1, 0); bw_put(b, 1, 0);
    bw_ue(b, 0); bw_ue(b, 0); bw_ue(b, 0);
    bw_put(b, 1, 0); bw_put(b, 2, 0);
    bw_se(b, e->pic_init_qp - 26);
    bw_put(b, 5, 0x1C);
    nal_finish(e);
}

/* H:4182-4373 encode_slice_header, restricted to KEY / P frames without long-term references */
static void write_slice_header(h264o_enc_t *e, int key)
{
    bitw_t *b = &e->bw;
    nal_begin(e, key ? 0x65 : 0x61);
    bw_ue(b, (uint32_t)e->slice_start_num);         /* first_mb_in_slice */
    bw_ue(b, (uint32_t)e->slice_type);
    bw_ue(b, 0);                                    /* pps id */
    bw_put(b, 5, (uint32_t)(e->frame_num & 31));
    if (key) bw_ue(b, (uint32_t)e->next_idr_pic_id);
    if (e->slice_type == SLICE_P) bw_put(b, 2, 0);  /* num_ref_idx_active_override_flag, ref_pic_list_modification_flag_l0 */
    if (key) bw_put(b, 2, 0);                       /* no_output_of_prior_pics_flag, long_term_reference_flag */
    else bw_put(b, 1, 0);                           /* adaptive_ref_pic_marking_mode_flag */
    bw_se(b, e->prev_qp - e->pic_init_qp);
    bw_ue(b, (uint32_t)(e->par.slices > 1 ? (e->no_deblock ? 1 : 2) : e->no_deblock));      /* H:4315-4323: idc 2 with row-band slices */
    if (e->no_deblock != 1) bw_put(b, 2, 3);
}

/* ------------------------------------------------------------------ rate control (frame level) */

static uint32_t mul32x32shr16(uint32_t x, uint32_t y)       /* H:3420-3425 */
{
    return (x >> 16)*(y & 0xFFFFu) + x*(y >> 16) + ((y & 0xFFFFu)*(x & 0xFFFFu) >> 16);
}

static uint32_t div_q16(uint32_t numer, uint32_t denum)     /* H:3430-3440 */
{
    unsigned f = 1u << __builtin_clz(denum);
    do
    {
        denum = denum*f >> 16;
        numer = mul32x32shr16(numer, f);
        f = ((1 << 17) - denum);
    } while (denum != 0xffff);
    return numer;
}

/* H:5924-6070 rc_frame_start (no long-term references) */
static void rc_frame_start(h264o_enc_t *e, int is_intra)
{
    unsigned np = (unsigned)(e->par.gop - 1u) < 63u ? (unsigned)(e->par.gop - 1u) : 63u;
    int nmb = e->nmb, qp = -1, add_bits, bit_budget = e->desired_frame_bytes*8, nominal_p, gop_bits, stationary;
    uint32_t peak_q16;
    do
    {
        qp++;
        gop_bits = (int)(k_bits_per_mb[0][qp]*np + k_bits_per_mb[1][qp]);
    } while (gop_bits*nmb > (int)(np + 1)*e->desired_frame_bytes*8 && qp < 40);
    peak_q16 = div_q16((uint32_t)k_bits_per_mb[1][qp] << 16, (uint32_t)k_bits_per_mb[0][qp] << 16);
    if (np)
    {
        uint32_t ratio = div_q16((np + 1) << 16, (np << 16) + peak_q16);
        nominal_p = (int)mul32x32shr16((uint32_t)(e->desired_frame_bytes*8), ratio);
    } else
        nominal_p = 0;
    stationary = imin(e->par.vbv_size_bytes*8 >> 4, e->desired_frame_bytes*8);
    if (is_intra)
        add_bits = (int)mul32x32shr16((uint32_t)nominal_p, peak_q16) - bit_budget;
    else
    {
        add_bits = nominal_p - bit_budget;
        if (e->par.vbv_size_bytes) add_bits += (e->rc.vbv_target_level - e->rc.vbv_bits) >> 4;
    }
    if (e->par.vbv_size_bytes) add_bits = imin(add_bits, (e->par.vbv_size_bytes*8*7 >> 3) - e->rc.vbv_bits);
    bit_budget += add_bits;
    bit_budget = imin(bit_budget, e->desired_frame_bytes*8*16);
    bit_budget = imax(bit_budget, e->desired_frame_bytes*8 >> 2);
    if (is_intra) e->rc.vbv_target_level = e->rc.vbv_bits + bit_budget - e->desired_frame_bytes*8;
    e->rc.vbv_target_level -= e->desired_frame_bytes*8 - nominal_p;
    e->rc.vbv_target_level = imax(e->rc.vbv_target_level, stationary);
    e->rc.bit_budget = bit_budget;
    {
        const uint16_t *bits = k_bits_per_mb[!!is_intra];
        for (qp = 0; qp < 42 - 1; qp++)
            if (bits[qp]*nmb < bit_budget) break;
    }
    qp += 10;
    qp += e->rc.dqp_smooth;
    if (e->prev_qp > qp + 1) qp = (e->prev_qp + qp + 1)/2;
    qp = imin(qp, e->qp_max); qp = imax(qp, e->qp_min); qp = imin(qp, 51);     /* H:5841-5843 */
    e->qp = qp;
    build_qdat(e->qdat, qp, e->slice_type == SLICE_P);
    e->rc.qp_smooth = qp << 8;
    e->prev_qp = qp;
}

/* H:6075-6141 rc_frame_end (no long-term references, no stuffing, overflow ignored as encode_app configures it) */
static void rc_frame_end(h264o_enc_t *e, int intra, int all_skipped)
{
    if (!all_skipped)
    {
        int qp, nmb = e->nmb;
        for (qp = 0; qp != 41 && k_bits_per_mb[intra][qp]*nmb > (int)e->out_pos*8 - 32; qp++) {}
        qp += 10;
        if ((e->rc.qp_smooth >> 8) - e->rc.dqp_smooth < qp - 1) e->rc.dqp_smooth--;
        else if ((e->rc.qp_smooth >> 8) - e->rc.dqp_smooth > qp + 1) e->rc.dqp_smooth++;
        if (intra) e->rc.max_dqp = e->rc.dqp_smooth;
        else e->rc.max_dqp = imax(e->rc.max_dqp, (e->rc.qp_smooth >> 8) - qp);
    }
    e->rc.vbv_bits += (int)e->out_pos*8 - e->desired_frame_bytes*8;
    if (e->par.vbv_size_bytes)
    {
        if (e->rc.vbv_bits < 0) e->rc.vbv_bits = 0;
        if (e->rc.vbv_bits > e->par.vbv_size_bytes*8) e->rc.vbv_bits = e->par.vbv_size_bytes*8;
    } else
        e->rc.vbv_bits = 0;
}

/* ------------------------------------------------------------------ frame / sequence */

/* H:2232-2248 h264e_copy_borders */
static void extend_borders(uint8_t *p, int w, int h, int stride, int guard)
{
    int y;
    for (y = 0; y < h; y++)
    {
        memset(p + y*stride - guard, p[y*stride], (size_t)guard);
        memset(p + y*stride + w, p[y*stride + w - 1], (size_t)guard);
    }
    for (y = 1; y <= guard; y++)
    {
        memcpy(p - y*stride - guard, p - guard, (size_t)(w + 2*guard));
        memcpy(p + (h - 1 + y)*stride - guard, p + (h - 1)*stride - guard, (size_t)(w + 2*guard));
    }
}

h264o_enc_t *h264o_open(const h264o_param_t *par)
{
    h264o_enc_t *e;
    size_t plane[3], off = 0;
    int c;
    if (!par || par->width <= 0 || par->height <= 0 || ((par->width | par->height) & 1)) return NULL;
    e = (h264o_enc_t *)calloc(1, sizeof(*e));
    if (!e) return NULL;
    e->par = *par;
    e->nmbx = (par->width + 15) >> 4; e->nmby = (par->height + 15) >> 4; e->nmb = e->nmbx*e->nmby;
    e->w = e->nmbx*16; e->h = e->nmby*16;
    e->cropping = !!((par->width | par->height) & 15);
    e->mv_limit.x0 = e->mv_limit.y0 = -14*4;
    e->mv_limit.x1 = (e->w - 2)*4; e->mv_limit.y1 = (e->h - 2)*4;
    e->mv_qlimit.x0 = e->mv_qlimit.y0 = -14*4 + 16;
    e->mv_qlimit.x1 = e->mv_limit.x1 - 16; e->mv_qlimit.y1 = e->mv_limit.y1 - 16;
    for (c = 0; c < 3; c++)
    {
        int s = (e->w + 2*GUARD) >> (c ? 1 : 0), hh = (e->h + 2*GUARD) >> (c ? 1 : 0);
        e->stride[c] = s;
        plane[c] = (size_t)s*hh;
    }
    e->mem = (uint8_t *)calloc(2*(plane[0] + plane[1] + plane[2]) + 64, 1);
    for (c = 0; c < 3; c++)
    {
        int g = GUARD >> (c ? 1 : 0);
        e->ref[c] = e->mem + off + (size_t)g*e->stride[c] + g; off += plane[c];
        e->dec[c] = e->mem + off + (size_t)g*e->stride[c] + g; off += plane[c];
    }
    e->mv_top = (mv32 *)calloc((size_t)e->nmbx*4 + 8, sizeof(mv32));
    e->nnz_top = (uint8_t *)calloc((size_t)e->nmbx*8 + 8, 1);
    e->i4_top = (int8_t *)calloc((size_t)e->nmbx*4 + 4, 1);
    e->pix_top = (uint8_t *)calloc((size_t)e->nmbx*32 + 64, 1);
    e->df_qp = (uint8_t *)calloc((size_t)e->nmbx + 1, 1);
    e->df_type = (int8_t *)calloc((size_t)e->nmbx + 1, 1);
    e->df_nz_top = (uint8_t *)calloc((size_t)e->nmbx + 1, 1);
    e->rbsp_cap = (size_t)e->nmb*768 + 4096;
    e->rbsp = (uint8_t *)malloc(e->rbsp_cap);
    e->out_cap = e->rbsp_cap*3/2 + 4096;
    e->out = (uint8_t *)malloc(e->out_cap);
    e->trace = (h264o_mbtrace_t *)calloc((size_t)e->nmb, sizeof(h264o_mbtrace_t));
    e->speed = par->speed;
    if (par->kbps)
    {
        e->desired_frame_bytes = par->kbps*1000/8/30;           /* T:596-600 */
        e->qp_min = 10; e->qp_max = 50;
    } else
    {
        e->qp_min = e->qp_max = par->qp;
    }
    if (!e->qp_max || e->qp_max > 51) e->qp_max = 51;            /* H:6707-6715 */
    if (!e->qp_min || e->qp_min < 10) e->qp_min = 10;
    return e;
}

void h264o_close(h264o_enc_t *e)
{
    if (!e) return;
    free(e->mem); free(e->mv_top); free(e->nnz_top); free(e->i4_top); free(e->pix_top);
    free(e->df_qp); free(e->df_type); free(e->df_nz_top); free(e->rbsp); free(e->out); free(e->trace);
    free(e);
}

int h264o_get_qp(const h264o_enc_t *e) { return e->qp; }     /* QP of the last encoded frame (rate control tests) */

void h264o_get_chain(const h264o_enc_t *e, h264o_chain_t *c)
{
    c->mv_clusters[0] = e->clusters[0]; c->mv_clusters[1] = e->clusters[1]; c->next_idr_pic_id = e->next_idr_pic_id;
}

void h264o_set_chain(h264o_enc_t *e, const h264o_chain_t *c)
{
    e->clusters[0] = c->mv_clusters[0]; e->clusters[1] = c->mv_clusters[1]; e->next_idr_pic_id = c->next_idr_pic_id & 1;
}

/* H:6898-6913 H264E_set_vbv_state: new VBV size (SPS level at the next key frame, the controller's limits) and, when >= 0, fullness */
void h264o_set_vbv_state(h264o_enc_t *e, int vbv_size_bytes, int vbv_fullness_bytes)
{
    e->par.vbv_size_bytes = vbv_size_bytes;
    if (vbv_fullness_bytes >= 0)
    {
        e->rc.vbv_bits = vbv_fullness_bytes*8;
        e->rc.vbv_target_level = e->rc.vbv_bits;
    }
}

const h264o_mbtrace_t *h264o_get_trace(const h264o_enc_t *e, int *nmb) { if (nmb) *nmb = e->nmb; return e->trace; }

void h264o_get_recon(const h264o_enc_t *e, uint8_t *dst, int *cw, int *ch)
{
    int c, y;
    if (cw) *cw = e->w;
    if (ch) *ch = e->h;
    if (!dst) return;
    for (c = 0; c < 3; c++)
    {
        int w = e->w >> (c ? 1 : 0), h = e->h >> (c ? 1 : 0);
        for (y = 0; y < h; y++, dst += w) memcpy(dst, e->ref[c] + (size_t)y*e->stride[c], (size_t)w);
    }
}

/* H:6654-6861 H264E_encode + H:6477-6626 H264E_encode_one + H:6409-6461 encode_slice */
int h264o_encode(h264o_enc_t *e, const uint8_t *const yuv[3], const int stride[3], uint8_t **out, int *out_bytes)
{
    int key = e->frame_num == 0, c, x, y;
    for (c = 0; c < 3; c++) { e->in[c] = yuv[c]; e->in_stride[c] = stride[c]; }
    e->out_pos = 0;
    e->no_deblock = (e->speed == 8 || e->speed == 10);
    if (key)
    {
        e->pic_init_qp = imax(imin(30, e->qp_max), e->qp_min);
        e->next_idr_pic_id ^= 1;
        e->frame_num = 0;
        write_sps(e);
        write_pps(e);
    }
    e->slice_type = key ? SLICE_I : SLICE_P;
    rc_frame_start(e, key);

    if (e->par.vbv_size_bytes && e->rc.vbv_bits - e->desired_frame_bytes*8 > e->par.vbv_size_bytes*8)
    {
        /* H:6497-6510 "encode transparent frame on VBV overflow" (reachable only right after H264E_set_vbv_state: rc_frame_end clamps
         * the fullness to the VBV size): slice header, one skip run over the whole picture -- written for key frames too --, and the
         * reference picture as the reconstruction */
        e->slice_start_row = e->slice_start_num = 0;
        write_slice_header(e, key);
        e->skip_run = e->nmb;
        bw_ue(&e->bw, (uint32_t)e->skip_run);
        nal_finish(e);
        for (c = 0; c < 3; c++)
            for (y = 0; y < (e->nmby*16 >> (c ? 1 : 0)); y++)
                memcpy(e->dec[c] + (size_t)y*e->stride[c], e->ref[c] + (size_t)y*e->stride[c], (size_t)(e->nmbx*16 >> (c ? 1 : 0)));
        rc_frame_end(e, key, 1);
    } else
    {
        /* one slice, or N row bands (H:6511-6574): every band is a slice of its own -- contexts, availability, skip run and
         * the mv_clusters state restart at its first macroblock (the reference encodes each band with a COPY of the encoder and
         * throws the copy away, so the parent's mv_clusters never move, H:6526) */
        const int nsl = e->par.slices > 1 ? e->par.slices : 1;
        const mv32 c0 = e->clusters[0], c1 = e->clusters[1];
        int band, row0 = 0;
        for (band = 0; band < nsl; band++)
        {
            const int row1 = row0 + (e->nmby - row0)/(nsl - band);
            e->slice_start_row = row0; e->slice_start_num = row0*e->nmbx;
            write_slice_header(e, key);
            e->skip_run = 0;
            memset(e->i4_top, -1, (size_t)e->nmbx*4); memset(e->i4_left, -1, 4);
            memset(e->nnz_top, NNZ_NA, (size_t)e->nmbx*8); memset(e->nnz_left, NNZ_NA, 8);
            for (y = row0; y < row1; y++)
            {
                for (x = 0; x < e->nmbx; x++) encode_mb(e, x, y);
                memset(e->nnz_left, NNZ_NA, 8);
                memset(e->i4_left, -1, 4);
            }
            if (e->skip_run) bw_ue(&e->bw, (uint32_t)e->skip_run);
            nal_finish(e);
            if (nsl > 1) { e->clusters[0] = c0; e->clusters[1] = c1; e->skip_run = 0; }
            row0 = row1;
        }
        e->slice_start_row = e->slice_start_num = 0;
        rc_frame_end(e, key, e->par.slices > 1 ? 0 : e->skip_run == e->nmb);       /* the parent's skip_run stays 0 in the threads build (H:6596) */
    }

    for (c = 0; c < 3; c++)
    {
        uint8_t *t = e->ref[c]; e->ref[c] = e->dec[c]; e->dec[c] = t;      /* H:3580-3596 */
        extend_borders(e->ref[c], e->w >> (c ? 1 : 0), e->h >> (c ? 1 : 0), e->stride[c], GUARD >> (c ? 1 : 0));
    }
    if (++e->frame_num >= e->par.gop && e->par.gop) e->frame_num = 0;
    e->frames_done++;
    *out = e->out;
    *out_bytes = (int)e->out_pos;
    return 0;
}

long h264o_encode_clip(const h264o_param_t *par, const uint8_t *clip, int nframes, uint8_t *out, size_t cap, int *frame_bytes)
{
    h264o_enc_t *e = h264o_open(par);
    size_t pos = 0, fsz = (size_t)par->width*par->height*3/2;
    int i;
    if (!e) return -1;
    for (i = 0; i < nframes; i++)
    {
        const uint8_t *f = clip + i*fsz, *yuv[3];
        int stride[3], n;
        uint8_t *p;
        yuv[0] = f; yuv[1] = f + (size_t)par->width*par->height; yuv[2] = yuv[1] + (size_t)(par->width/2)*(par->height/2);
        stride[0] = par->width; stride[1] = stride[2] = par->width/2;
        h264o_encode(e, yuv, stride, &p, &n);
        if (pos + (size_t)n > cap) { h264o_close(e); return -1; }
        memcpy(out + pos, p, (size_t)n);
        pos += (size_t)n;
        if (frame_bytes) frame_bytes[i] = n;
    }
    h264o_close(e);
    return (long)pos;
}
