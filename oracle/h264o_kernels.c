/*
 * h264o_kernels.c -- pixel and bit-level kernels of the CPU oracle.
 * TEST INFRASTRUCTURE (see h264o_internal.h).  "H:n" = /root/reference/src/h264-lab.h:n.
 */
#include "h264o_internal.h"
#include "h264o_tables.h"

static inline int iabs(int x) { return x < 0 ? -x : x; }
static inline int clip255(int x) { return x < 0 ? 0 : x > 255 ? 255 : x; }
static inline int clip3(int lo, int hi, int x) { return x < lo ? lo : x > hi ? hi : x; }

/* ------------------------------------------------------------------ bits */

void bw_init(bitw_t *b, uint8_t *buf, size_t cap) { b->buf = buf; b->cap = cap; b->acc = 0; b->nacc = 0; b->pos = 0; }

/* H:2688-2702 h264e_bs_put_bits: MSB-first, at most 32 bits per call */
void bw_put(bitw_t *b, int n, uint32_t v)
{
    if (!n) return;
    b->acc = (b->acc << n) | (uint64_t)(n == 32 ? v : (v & ((1u << n) - 1)));
    b->nacc += n;
    while (b->nacc >= 8)
    {
        b->nacc -= 8;
        if (b->pos < b->cap) b->buf[b->pos] = (uint8_t)(b->acc >> b->nacc);
        b->pos++;
    }
}

size_t bw_bits(const bitw_t *b) { return b->pos*8 + b->nacc; }

void bw_flush(bitw_t *b) { if (b->nacc) bw_put(b, 8 - b->nacc, 0); }

/* H:2738-2747 Exp-Golomb ue(v): 2*size-1 bits holding v+1 */
void bw_ue(bitw_t *b, uint32_t v)
{
    int size = 0;
    uint32_t t = v + 1;
    do size++; while (t >>= 1);
    bw_put(b, 2*size - 1, v + 1);
}

/* H:2760-2765 se(v): v>0 -> 2v-1, v<=0 -> -2v */
void bw_se(bitw_t *b, int v) { bw_ue(b, v > 0 ? 2*v - 1 : -2*v); }

/* H:3402-3415 */
int bits_ue(int v) { return 2*(32 - __builtin_clz((unsigned)v + 1)) - 1; }
int bits_se(int v) { return bits_ue(v > 0 ? 2*v - 1 : -2*v); }

/* ------------------------------------------------------------------ SAD */

/* H:2162-2176 sad_block */
int sad_wh(const uint8_t *a, int as, const uint8_t *b, int bs, int w, int h)
{
    int x, y, s = 0;
    for (y = 0; y < h; y++, a += as, b += bs)
        for (x = 0; x < w; x++) s += iabs(a[x] - b[x]);
    return s;
}

/* H:2178-2187 h264e_sad_mb_unlaign_8x8: four 8x8 quadrant SADs and their sum */
int sad_16x16_q(const uint8_t *a, int as, const uint8_t *b, int bs, int sad4[4])
{
    sad4[0] = sad_wh(a, as, b, bs, 8, 8);
    sad4[1] = sad_wh(a + 8, as, b + 8, bs, 8, 8);
    sad4[2] = sad_wh(a + 8*as, as, b + 8*bs, bs, 8, 8);
    sad4[3] = sad_wh(a + 8*as + 8, as, b + 8*bs + 8, bs, 8, 8);
    return sad4[0] + sad4[1] + sad4[2] + sad4[3];
}

/* ------------------------------------------------------------------ inter prediction */

static inline int tap6(const uint8_t *p, int s) { return p[-2*s] - 5*p[-s] + 20*p[0] + 20*p[s] - 5*p[2*s] + p[3*s]; }
static inline int hp_h(const uint8_t *p) { return clip255((tap6(p, 1) + 16) >> 5); }            /* H:2029-2039 */
static inline int hp_v(const uint8_t *p, int s) { return clip255((tap6(p, s) + 16) >> 5); }     /* H:2041-2051 */
static inline int hp_d(const uint8_t *p, int s)                                                 /* H:1990-2027 */
{
    int k, t[6];
    for (k = 0; k < 6; k++) t[k] = tap6(p + (k - 2)*s, 1);
    return clip255((t[0] - 5*t[1] + 20*t[2] + 20*t[3] - 5*t[4] + t[5] + 512) >> 10);
}

/*
 * H:4905-4910 interpolate_luma + H:2079-2131 h264e_qpel_interpolate_luma: the standard
 * H.264 quarter-sample luma interpolation (half samples b,h,j; quarter samples are rounded
 * averages of the two nearest integer/half samples).  mv is absolute, in quarter samples.
 */
void interp_luma(const uint8_t *ref, int stride, int mx, int my, int w, int h, uint8_t *dst)
{
    int fx = mx & 3, fy = my & 3, x, y;
    const uint8_t *base = ref + (my >> 2)*stride + (mx >> 2);
    for (y = 0; y < h; y++)
    {
        for (x = 0; x < w; x++)
        {
            const uint8_t *p = base + y*stride + x;
            int v;
            switch (fx + 4*fy)
            {
            default:
            case 0:  v = p[0]; break;
            case 1:  v = (p[0] + hp_h(p) + 1) >> 1; break;
            case 2:  v = hp_h(p); break;
            case 3:  v = (p[1] + hp_h(p) + 1) >> 1; break;
            case 4:  v = (p[0] + hp_v(p, stride) + 1) >> 1; break;
            case 5:  v = (hp_h(p) + hp_v(p, stride) + 1) >> 1; break;
            case 6:  v = (hp_h(p) + hp_d(p, stride) + 1) >> 1; break;
            case 7:  v = (hp_h(p) + hp_v(p + 1, stride) + 1) >> 1; break;
            case 8:  v = hp_v(p, stride); break;
            case 9:  v = (hp_v(p, stride) + hp_d(p, stride) + 1) >> 1; break;
            case 10: v = hp_d(p, stride); break;
            case 11: v = (hp_v(p + 1, stride) + hp_d(p, stride) + 1) >> 1; break;
            case 12: v = (p[stride] + hp_v(p, stride) + 1) >> 1; break;
            case 13: v = (hp_h(p + stride) + hp_v(p, stride) + 1) >> 1; break;
            case 14: v = (hp_h(p + stride) + hp_d(p, stride) + 1) >> 1; break;
            case 15: v = (hp_h(p + stride) + hp_v(p + 1, stride) + 1) >> 1; break;
            }
            dst[y*16 + x] = (uint8_t)v;
        }
    }
}

/* H:2133-2160 h264e_qpel_interpolate_chroma: 1/8-sample bilinear; mv is the LUMA mv (absolute, qpel) */
void interp_chroma(const uint8_t *ref, int stride, int mx, int my, int w, int h, uint8_t *dst)
{
    int dx = mx & 7, dy = my & 7, x, y;
    const uint8_t *p = ref + (my >> 3)*stride + (mx >> 3);
    int a = (8 - dx)*(8 - dy), b = dx*(8 - dy), c = (8 - dx)*dy, d = dx*dy;
    for (y = 0; y < h; y++, p += stride)
        for (x = 0; x < w; x++)
            dst[y*16 + x] = (dx | dy) ? (uint8_t)((a*p[x] + b*p[x + 1] + c*p[stride + x] + d*p[stride + x + 1] + 32) >> 6) : p[x];
}

/* H:2065-2077 */
void avg_wh(const uint8_t *a, const uint8_t *b, uint8_t *d, int w, int h)
{
    int x, y;
    for (y = 0; y < h; y++)
        for (x = 0; x < w; x++) d[y*16 + x] = (uint8_t)((a[y*16 + x] + b[y*16 + x] + 1) >> 1);
}

/* ------------------------------------------------------------------ intra prediction */

/* H:1625-1651 intra_predict_dc: mean of the available edges, 128 when none (NULL = unavailable) */
static int dc_pred(const uint8_t *left, const uint8_t *top, int n)
{
    int i, s = 0, cnt = 0;
    if (left) { for (i = 0; i < n; i++) s += left[i]; cnt += n; }
    if (top)  { for (i = 0; i < n; i++) s += top[i];  cnt += n; }
    if (!cnt) return 128;
    return (s + cnt/2)/cnt;     /* cnt is n or 2n, both powers of two: identical to the shifts of the reference */
}

/* H:1677-1714: mode 0 vertical, 1 horizontal, 2 DC */
void pred16(uint8_t *dst, const uint8_t *left, const uint8_t *top, int mode)
{
    int x, y, dc = mode == 2 ? dc_pred(left, top, 16) : 0;
    for (y = 0; y < 16; y++)
        for (x = 0; x < 16; x++)
            dst[y*16 + x] = (uint8_t)(mode == 0 ? top[x] : mode == 1 ? left[y] : dc);
}

/*
 * H:1716-1781: 8x8 U (columns 0..7) and V (columns 8..15) side by side, stride 16.
 * left/top hold U then V (8 + 8).  mode follows the LUMA 16x16 numbering (0 V, 1 H, 2 DC).
 */
void pred_chroma(uint8_t *dst, const uint8_t *left, const uint8_t *top, int mode)
{
    int c, x, y;
    for (c = 0; c < 2; c++)
    {
        const uint8_t *l = left ? left + 8*c : NULL, *t = top ? top + 8*c : NULL;
        uint8_t *d = dst + 8*c;
        int dc[4];
        if (mode == 2)
        {
            /* per 4x4 quadrant, H.264 8.3.4.1-3 (same rule set as H:1755-1764) */
            dc[0] = dc_pred(l, t, 4);
            dc[1] = t ? dc_pred(NULL, t + 4, 4) : dc_pred(l, NULL, 4);
            dc[2] = l ? dc_pred(l + 4, NULL, 4) : dc_pred(NULL, t, 4);
            dc[3] = dc_pred(l ? l + 4 : NULL, t ? t + 4 : NULL, 4);
        }
        for (y = 0; y < 8; y++)
            for (x = 0; x < 8; x++)
                d[y*16 + x] = (uint8_t)(mode == 0 ? t[x] : mode == 1 ? l[y] : dc[(y >> 2)*2 + (x >> 2)]);
    }
}

/*
 * H:1810-1962 h264e_intra_choose_4x4.  Try DC first, then V,DDL,VL (need top), H,HU (need left),
 * DDR,HD,VR (need top, left and top-left), keep the first strict minimum of SAD (+penalty when
 * the mode differs from the predicted one).  top8 = U0..U7, left4 = L0..L3.
 * Writes the winning prediction to pred (stride 16), returns the mode, *psad = its cost.
 */
int i4_choose(const uint8_t *in, uint8_t *pred, int avail, const uint8_t *top8, const uint8_t *left4, int tl,
              int mpred, int penalty, int *psad)
{
    static const uint8_t order[9] = { 2, 0, 3, 7, 1, 8, 4, 6, 5 };
    uint8_t t[8], p[16];
    const uint8_t *l = left4;
    int k, x, y, best = 2, best_sad = 0;
    if (avail & AV_T)
    {
        for (k = 0; k < 8; k++) t[k] = top8[k];
        if (!(avail & AV_TR)) for (k = 4; k < 8; k++) t[k] = t[3];      /* H:1850-1853 */
    }
    for (k = 0; k < 9; k++)
    {
        int m = order[k], sad = 0;
        if ((m == 0 || m == 3 || m == 7) && !(avail & AV_T)) continue;
        if ((m == 1 || m == 8) && !(avail & AV_L)) continue;
        if ((m == 4 || m == 5 || m == 6) && (avail & (AV_T | AV_L | AV_TL)) != (AV_T | AV_L | AV_TL)) continue;
        for (y = 0; y < 4; y++)
        {
            for (x = 0; x < 4; x++)
            {
                int v, z;
                /* e(i): edge sample i steps clockwise from the corner (i > 0 top, i < 0 left, 0 = top-left) */
#define E(i) ((i) == 0 ? tl : (i) > 0 ? t[(i) - 1] : l[-(i) - 1])
                switch (m)
                {
                case 0: v = t[x]; break;
                case 1: v = l[y]; break;
                default:
                case 2: v = dc_pred((avail & AV_L) ? l : NULL, (avail & AV_T) ? t : NULL, 4); break;
                case 3: v = (x + y == 6) ? (t[6] + 3*t[7] + 2) >> 2 : (t[x + y] + 2*t[x + y + 1] + t[x + y + 2] + 2) >> 2; break;
                case 4: z = x - y; v = (E(z - 1) + 2*E(z) + E(z + 1) + 2) >> 2; break;
                case 5: z = 2*x - y;
                    if (z >= 0 && !(z & 1)) v = (E(x - (y >> 1)) + E(x - (y >> 1) + 1) + 1) >> 1;
                    else if (z >= 0)        v = (E(x - (y >> 1) - 1) + 2*E(x - (y >> 1)) + E(x - (y >> 1) + 1) + 2) >> 2;
                    else if (z == -1)       v = (E(-1) + 2*E(0) + E(1) + 2) >> 2;
                    else                    v = (E(-y) + 2*E(-(y - 1)) + E(-(y - 2)) + 2) >> 2;
                    break;
                case 6: z = 2*y - x;
                    if (z >= 0 && !(z & 1)) v = (E(-(y - (x >> 1))) + E(-(y - (x >> 1) + 1)) + 1) >> 1;
                    else if (z >= 0)        v = (E(-(y - (x >> 1) - 1)) + 2*E(-(y - (x >> 1))) + E(-(y - (x >> 1) + 1)) + 2) >> 2;
                    else if (z == -1)       v = (E(-1) + 2*E(0) + E(1) + 2) >> 2;
                    else                    v = (E(x) + 2*E(x - 1) + E(x - 2) + 2) >> 2;
                    break;
                case 7: v = (y & 1) ? (t[x + (y >> 1)] + 2*t[x + (y >> 1) + 1] + t[x + (y >> 1) + 2] + 2) >> 2
                                    : (t[x + (y >> 1)] + t[x + (y >> 1) + 1] + 1) >> 1; break;
                case 8: z = x + 2*y;
                    if (z > 5)       v = l[3];
                    else if (z == 5) v = (l[2] + 3*l[3] + 2) >> 2;
                    else if (z & 1)  v = (l[y + (x >> 1)] + 2*l[y + (x >> 1) + 1] + l[y + (x >> 1) + 2] + 2) >> 2;
                    else             v = (l[y + (x >> 1)] + l[y + (x >> 1) + 1] + 1) >> 1;
                    break;
                }
#undef E
                p[y*4 + x] = (uint8_t)v;
                sad += iabs(in[y*16 + x] - v);
            }
        }
        if (m != mpred) sad += penalty;
        if (k == 0 || sad < best_sad)
        {
            best_sad = sad;
            best = m;
            for (y = 0; y < 4; y++) for (x = 0; x < 4; x++) pred[y*16 + x] = p[y*4 + x];
        }
    }
    *psad = best_sad;
    return best;
}

/* ------------------------------------------------------------------ transform / quant */

/* H:2374-2409 forward 4x4 core transform of (inp - pred); out index = 4*k_h + k_v */
static void fwd4x4(const uint8_t *inp, int is, const uint8_t *pred, int16_t *out)
{
    int x, k, tmp[4][4];            /* tmp[x][k_v] */
    for (x = 0; x < 4; x++)
    {
        int f0 = inp[x] - pred[x], f1 = inp[is + x] - pred[16 + x];
        int f2 = inp[2*is + x] - pred[32 + x], f3 = inp[3*is + x] - pred[48 + x];
        int t0 = f0 + f3, t1 = f0 - f3, t2 = f1 + f2, t3 = f1 - f2;
        tmp[x][0] = t0 + t2; tmp[x][1] = t1*2 + t3; tmp[x][2] = t0 - t2; tmp[x][3] = t1 - t3*2;
    }
    for (k = 0; k < 4; k++)
    {
        int d0 = tmp[0][k], d1 = tmp[1][k], d2 = tmp[2][k], d3 = tmp[3][k];
        int t0 = d0 + d3, t1 = d0 - d3, t2 = d1 + d2, t3 = d1 - d2;
        out[k] = (int16_t)(t0 + t2); out[4 + k] = (int16_t)(t1*2 + t3);
        out[8 + k] = (int16_t)(t0 - t2); out[12 + k] = (int16_t)(t1 - t3*2);
    }
}

/* H:2436-2489 inverse transform, horizontal pass first; in index = 4*k_h + k_v, out = raster 4*y + x */
static void inv4x4(int16_t *c)
{
    int i;
    int16_t tmp[16];
    for (i = 0; i < 4; i++)         /* i = k_v */
    {
        int d0 = c[i], d1 = c[i + 4], d2 = c[i + 8], d3 = c[i + 12];
        int e0 = d0 + d2, e1 = d0 - d2, e2 = (d1 >> 1) - d3, e3 = d1 + (d3 >> 1);
        tmp[4*i + 0] = (int16_t)(e0 + e3); tmp[4*i + 1] = (int16_t)(e1 + e2);
        tmp[4*i + 2] = (int16_t)(e1 - e2); tmp[4*i + 3] = (int16_t)(e0 - e3);
    }
    for (i = 0; i < 4; i++)         /* i = x */
    {
        int f0 = tmp[i], f1 = tmp[i + 4], f2 = tmp[i + 8], f3 = tmp[i + 12];
        int g0 = f0 + f2, g1 = f0 - f2, g2 = (f1 >> 1) - f3, g3 = f1 + (f3 >> 1);
        c[i]      = (int16_t)((g0 + g3 + 32) >> 6); c[i + 4]  = (int16_t)((g1 + g2 + 32) >> 6);
        c[i + 8]  = (int16_t)((g1 - g2 + 32) >> 6); c[i + 12] = (int16_t)((g0 - g3 + 32) >> 6);
    }
}

/* H:2491-2502 is_zero: every coefficient from i0 on lies inside [-thr, thr] */
static int small_block(const int16_t *c, int i0, const uint16_t *thr)
{
    int i;
    for (i = i0; i < 16; i++)
        if ((unsigned)(c[i] + thr[i & 7]) > (unsigned)2*thr[i & 7]) return 0;
    return 1;
}

/*
 * H:2619-2636 h264e_transform_sub_quant_dequant = transform (H:2599) + DC pick-off + dead zone
 * (zero_smallq H:2512) + quantize (H:2536).  Returns the non-zero block mask, first block in the
 * highest bit.  dc receives the unquantized DCs for mode I16 / chroma.
 */
int xform_quant(const uint8_t *inp, int is, const uint8_t *pred, int mode, qblk_t *q, int16_t *dc, const uint16_t *qdat)
{
    static const uint8_t cls[16] = { 0, 2, 0, 2, 2, 4, 2, 4, 0, 2, 0, 2, 2, 4, 2, 4 };   /* H:2366 */
    int n = mode >> 1, i0 = mode & 1, bx, by, b, i, zmask = 0, nzmask = 0;
    for (by = 0; by < n; by++)
        for (bx = 0; bx < n; bx++)
            fwd4x4(inp + 4*by*is + 4*bx, is, pred + 4*by*16 + 4*bx, q[by*n + bx].dq);
    if (i0)
        for (b = 0; b < n*n; b++) dc[b] = q[b].dq[0];
    if (mode == QMODE_INTER || mode == QMODE_CHROMA)
    {
        for (b = 0; b < n*n; b++)
            if (small_block(q[b].dq, i0, qdat + QD_THR1)) zmask |= 1 << b;
        if (mode == QMODE_INTER)
        {
            static const uint8_t grp[4] = { 0, 2, 8, 10 };
            for (i = 0; i < 4; i++)
            {
                int g = grp[i], m = 0x33 << g;
                if ((~zmask & m) && small_block(q[g].dq, i0, qdat + QD_THR2) && small_block(q[g + 1].dq, i0, qdat + QD_THR2) &&
                    small_block(q[g + 4].dq, i0, qdat + QD_THR2) && small_block(q[g + 5].dq, i0, qdat + QD_THR2))
                    zmask |= m;
            }
        }
    }
    for (b = 0; b < n*n; b++)
    {
        int nz = 0;
        if (zmask & (1 << b))
        {
            memset(q[b].qv, 0, sizeof(q[b].qv));
        } else
        {
            for (i = i0; i < 16; i++)
            {
                int off = cls[i], rnd = qdat[QD_RND], v;
                if (q[b].dq[i] < 0) rnd = 0xFFFF - rnd;
                v = (q[b].dq[i]*qdat[off] + rnd) >> 16;
                nz |= v;
                q[b].qv[i] = (int16_t)v;
                q[b].dq[i] = (int16_t)(v*qdat[off + 1]);
            }
        }
        nzmask = (nzmask << 1) | (nz != 0);
    }
    return nzmask;
}

/* H:2269-2301 hadamar4_2d: 4x4 Hadamard, result transposed, every store truncated to int16 */
static void hadamard4(int16_t *x)
{
    int16_t tmp[16];
    int j, i;
    for (j = 0; j < 4; j++)
    {
        int a = x[j], b = x[4 + j], c = x[8 + j], d = x[12 + j];
        tmp[4*j + 0] = (int16_t)(a + b + c + d); tmp[4*j + 1] = (int16_t)(a + b - c - d);
        tmp[4*j + 2] = (int16_t)(a - b - c + d); tmp[4*j + 3] = (int16_t)(a - b + c - d);
    }
    for (i = 0; i < 4; i++)
    {
        int a = tmp[i], b = tmp[4 + i], c = tmp[8 + i], d = tmp[12 + i];
        x[i]     = (int16_t)(a + b + c + d); x[4 + i]  = (int16_t)(a + b - c - d);
        x[8 + i] = (int16_t)(a - b - c + d); x[12 + i] = (int16_t)(a - b + c - d);
    }
}

/* H:2308-2330 quant_dc */
static void quant_dc(int16_t *v, int16_t *lev, int quant, int n, int round_q18)
{
    int i;
    for (i = 0; i < n; i++)
    {
        int r = v[i] < 0 ? (1 << 18) - round_q18 : round_q18;
        lev[i] = v[i] = (int16_t)((v[i]*quant + r) >> 18);
    }
}

/* H:2344-2353 */
void quant_luma_dc(qblk_t *q, int16_t *dc, int16_t *lev, const uint16_t *qdat)
{
    int i;
    hadamard4(dc);
    quant_dc(dc, lev, (int16_t)qdat[0], 16, 0x20000);
    hadamard4(dc);
    for (i = 0; i < 16; i++) q[i].dq[0] = (int16_t)(dc[i]*(int16_t)(qdat[1] >> 2));
}

/* H:2355-2364 */
int quant_chroma_dc(qblk_t *q, int16_t *dc, int16_t *lev, const uint16_t *qdat)
{
    int i, k;
    for (k = 0; k < 2; k++)
    {
        int a = dc[0], b = dc[1], c = dc[2], d = dc[3];             /* H:2332-2342 hadamar2_2d */
        dc[0] = (int16_t)(a + b + c + d); dc[1] = (int16_t)(a - b + c - d);
        dc[2] = (int16_t)(a + b - c - d); dc[3] = (int16_t)(a - b - c + d);
        if (!k) quant_dc(dc, lev, (int16_t)(qdat[0] << 1), 4, 0xAAAA);
    }
    for (i = 0; i < 4; i++) q[i].dq[0] = (int16_t)(dc[i]*(int16_t)(qdat[1] >> 1));
    return !!(dc[0] | dc[1] | dc[2] | dc[3]);
}

/* H:2638-2681 h264e_transform_add: blocks whose mask bit (MSB first) is clear are plain copies */
void recon_blocks(uint8_t *out, int os, const uint8_t *pred, qblk_t *q, int side, uint32_t mask)
{
    int bx, by, x, y;
    for (by = 0; by < side; by++)
    {
        for (bx = 0; bx < side; bx++, q++, mask <<= 1)
        {
            uint8_t *o = out + 4*by*os + 4*bx;
            const uint8_t *p = pred + 4*by*16 + 4*bx;
            if (mask & 0x80000000u) inv4x4(q->dq);
            for (y = 0; y < 4; y++)
                for (x = 0; x < 4; x++)
                    o[y*os + x] = (mask & 0x80000000u) ? (uint8_t)clip255(q->dq[4*y + x] + p[16*y + x]) : p[16*y + x];
        }
    }
}

/* H:5822-5834 rc_rnd2thr: largest thr with thr*q <= 0x10000 - round */
static uint16_t rnd2thr(int round, int q)
{
    int b, thr = 0;
    for (b = 0x8000; b; b >>= 1)
        if ((thr | b)*q <= 0x10000 - round) thr |= b;
    return (uint16_t)thr;
}

/* H:5839-5912 rc_set_qp: quantizer tables for luma [0] and chroma [1] */
void build_qdat(uint16_t qdat[2][42], int qp, int p_slice)
{
    static const int16_t qc[6][6] = {        /* {quant, dequant} for position classes 0 / 2 / 1 */
        { 13107, 10, 8066, 13, 5243, 16 }, { 11916, 11, 7490, 14, 4660, 18 }, { 10082, 13, 6554, 16, 4194, 20 },
        {  9362, 14, 5825, 18, 3647, 23 }, {  8192, 16, 5243, 20, 3355, 25 }, {  7282, 18, 4559, 23, 2893, 29 } };
    int c, i, k, luma_qp = qp;
    for (c = 0; c < 2; c++)
    {
        uint16_t *d = qdat[c];
        int div6 = qp*86 >> 9, mod6 = qp - div6*6;
        for (i = 0; i < 3; i++)
        {
            d[2*i]     = (uint16_t)(qc[mod6][2*i] << 1 >> div6);
            d[2*i + 1] = (uint16_t)(qc[mod6][2*i + 1] << div6);
        }
        /* rounding tables are indexed by the CURRENT qp of the pass (luma qp, then chroma qp) */
        d[6] = p_slice ? k_rnd_inter[qp] : k_deadzonei[qp];
        d[7] = k_deadzonei[qp];
        d[8] = (uint16_t)(k_thr_inter[qp] - 0x7fff);
        d[9] = (uint16_t)(k_thr_inter2[qp] - 0x7fff);
        for (k = 0; k < 2; k++)
        {
            uint16_t *t = d + 10 + 8*k;
            int r = (k ? k_thr_inter2[qp] : k_thr_inter[qp]) - 0x7fff;
            t[0] = t[2] = rnd2thr(r, d[0]);
            t[1] = t[3] = t[4] = t[6] = rnd2thr(r, d[2]);
            t[5] = t[7] = rnd2thr(r, d[4]);
        }
        for (k = 0; k < 2; k++)
        {
            uint16_t *t = d + 26 + 8*k;
            t[0] = t[2] = d[k]; t[1] = t[3] = t[4] = t[6] = d[2 + k]; t[5] = t[7] = d[4 + k];
        }
        qp = k_qpc[qp];
    }
    (void)luma_qp;
}

/* ------------------------------------------------------------------ CAVLC */

/*
 * H:2775-2949 h264e_vlc_encode.  coef[first..first+maxn-1] are scanned in DECREASING index order
 * (the reference has no zig-zag).  nl / nt = raw left / top nnz context values
 * (NNZ_NA = unavailable, 17+17 selects the chroma-DC table).
 */
void cavlc_block(bitw_t *b, const int16_t *coef, int first, int maxn, int nctx, uint8_t *nnz_out)
{
    int lev[16], pos[16], total = 0, t1 = 0, i, k, tab, sl, zeros;
    for (i = maxn - 1; i >= 0; i--)
        if (coef[first + i]) { lev[total] = coef[first + i]; pos[total] = i; total++; }
    while (t1 < 3 && t1 < total && (lev[t1] == 1 || lev[t1] == -1)) t1++;
    *nnz_out = (uint8_t)total;

    if (nctx <= 34) nctx = (nctx + 1) >> 1;             /* H:2816-2823 */
    nctx &= 31;
    tab = nctx < 2 ? 0 : nctx < 4 ? 1 : nctx < 8 ? 2 : nctx < 17 ? 3 : 4;
    bw_put(b, k_coeff_token[tab][total][t1][0], k_coeff_token[tab][total][t1][1]);
    if (!total) return;

    for (i = 0; i < t1; i++) bw_put(b, 1, lev[i] < 0);
    sl = (total > 10 && t1 < 3) ? 1 : 0;
    for (i = t1; i < total; i++)
    {
        int a = lev[i] < 0 ? -lev[i] : lev[i];
        int code = 2*a - 2 + (lev[i] < 0), prefix, nsuf, suf;
        if (i == t1 && t1 < 3) code -= 2;
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
        zeros = pos[0] + 1 - total;
        if (maxn == 4) bw_put(b, k_total_zeros_cdc[total - 1][zeros][0], k_total_zeros_cdc[total - 1][zeros][1]);
        else           bw_put(b, k_total_zeros[total - 1][zeros][0], k_total_zeros[total - 1][zeros][1]);
        for (k = 0; k < total - 1 && zeros > 0; k++)
        {
            int run = pos[k] - pos[k + 1] - 1, zl = zeros > 7 ? 7 : zeros;
            bw_put(b, k_run_before[zl - 1][run][0], k_run_before[zl - 1][run][1]);
            zeros -= run;
        }
    }
}

/* ------------------------------------------------------------------ deblocking */

/* normal (bS < 4) luma edge sample, H:1251-1300 / H:1396-1447 */
static void df_luma_normal(uint8_t *p, int s, int alpha, int beta, int tc0)
{
    int p2 = p[-3*s], p1 = p[-2*s], p0 = p[-s], q0 = p[0], q1 = p[s], q2 = p[2*s];
    if (iabs(p0 - q0) < alpha && iabs(p1 - p0) < beta && iabs(q1 - q0) < beta)
    {
        int ap = iabs(p2 - p0) < beta, aq = iabs(q2 - q0) < beta;
        int tc = tc0 + ap + aq;
        int delta = clip3(-tc, tc, (((q0 - p0)*4) + (p1 - q1) + 4) >> 3);
        if (ap) p[-2*s] = (uint8_t)(p1 + clip3(-tc0, tc0, ((p2 + ((p0 + q0 + 1) >> 1)) >> 1) - p1));
        if (aq) p[s]    = (uint8_t)(q1 + clip3(-tc0, tc0, ((q2 + ((p0 + q0 + 1) >> 1)) >> 1) - q1));
        p[-s] = (uint8_t)clip255(p0 + delta);
        p[0]  = (uint8_t)clip255(q0 - delta);
    }
}

/* strong (bS = 4) luma edge sample, H:1302-1394 */
static void df_luma_strong(uint8_t *p, int s, int alpha, int beta)
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
static void df_chroma(uint8_t *p, int s, int alpha, int beta, int tc0, int bs)
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

/*
 * H:5642-5716 mb_deblock + H:1469-1545: filter the left/top and inner edges of one macroblock in
 * place.  bs[4*e + k] = strength of vertical edge e (x = 4e), rows 4k..4k+3; bs[16 + 4*e + k] =
 * horizontal edge e, columns 4k..4k+3.  An edge whose FIRST strength is 4 is strong-filtered over
 * all 16 samples (H:1517, H:1534).
 */
void deblock_mb(uint8_t *y, int ys, uint8_t *u, uint8_t *v, int cs, const uint8_t bs[32], int qp, int qp_left, int qp_top)
{
    int dir, e, k, i, c;
    for (dir = 0; dir < 2; dir++)
    {
        for (e = 0; e < 4; e++)
        {
            const uint8_t *s = bs + 16*dir + 4*e;
            int q = e ? qp : dir ? (qp_top + qp + 1) >> 1 : (qp_left + qp + 1) >> 1;
            int alpha = k_df_alpha[q], beta = k_df_beta[q];
            uint8_t *p = dir ? y + 4*e*ys : y + 4*e;
            int along = dir ? 1 : ys, across = dir ? ys : 1;
            if (!(s[0] | s[1] | s[2] | s[3])) continue;
            if (s[0] == 4)
            {
                for (i = 0; i < 16; i++) df_luma_strong(p + i*along, across, alpha, beta);
            } else if (alpha)
            {
                for (k = 0; k < 4; k++)
                    if (s[k])
                        for (i = 0; i < 4; i++)
                            df_luma_normal(p + (4*k + i)*along, across, alpha, beta, s[k] < 4 ? k_df_tc0[q][s[k] - 1] : beta);
            }
        }
    }
    for (c = 0; c < 2; c++)
    {
        uint8_t *pl = c ? v : u;
        int cq = k_qpc[qp], cql = k_qpc[qp_left], cqt = k_qpc[qp_top];
        for (dir = 0; dir < 2; dir++)
        {
            for (e = 0; e < 4; e += 2)
            {
                const uint8_t *s = bs + 16*dir + 4*e;
                int q = e ? cq : dir ? (cqt + cq + 1) >> 1 : (cql + cq + 1) >> 1;
                int alpha = k_df_alpha[q], beta = k_df_beta[q];
                uint8_t *p = dir ? pl + 2*e*cs : pl + 2*e;
                int along = dir ? 1 : cs, across = dir ? cs : 1;
                if (!(s[0] | s[1] | s[2] | s[3]) || !alpha) continue;
                for (i = 0; i < 8; i++)
                {
                    int st = s[i >> 1];
                    df_chroma(p + i*along, across, alpha, beta, (st && st < 4) ? k_df_tc0[q][st - 1] : 0, st);
                }
            }
        }
    }
}
