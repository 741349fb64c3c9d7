/*
 * h264e_host.c -- C host side of the MI355X encoder: the reference's public API (include/h264e_mi355x.h)
 * over the HIP C ABI (include/h264e_hip.h).
 *
 * What stays on the host, as in the reference: frame-type / GOP bookkeeping (h264-lab.h:6725-6811), SPS / PPS /
 * slice header syntax (h264-lab.h:4040-4333), NAL framing with emulation prevention (h264-lab.h:3926-4022),
 * frame-level rate control (h264-lab.h:5924-6141) and the quantizer tables (h264-lab.h:5839-5912).
 * What runs on the GPU: the whole macroblock loop (encode_slice -> mb_encode) and the slice splice.
 *
 * mv_clusters (h264-lab.h:766) is the one raster-serial state the GPU cannot carry: macroblocks are encoded
 * with a SPECULATED value and the host validates it afterwards: exact trajectory from the
 * per-macroblock records, comparison of the rounded start candidates actually consumed, re-encode of the frame
 * with the exact per-macroblock values when they differ.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <sched.h>
#include <pthread.h>
#include "../../include/h264e_mi355x.h"
#include "../../include/h264e_hip.h"

#define MAGIC 0x4D493335u   /* "MI35" */
#define SLICE_P 0
#define SLICE_I 2

static int g_device = -1;
static __thread char g_host_err[256];       /* per calling thread: encoders on different threads do not share error text */

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

const char *H264E_last_error(void) { return g_host_err[0] ? g_host_err : h264e_hip_last_error(); }
void H264E_set_device(int device) { g_device = device; }
int H264E_device_count(void) { return h264e_hip_device_count(); }

static int pick_device(void)
{
    const char *e;
    if (g_device >= 0) return g_device;
    e = getenv("H264E_DEVICE");
    return e ? atoi(e) : 0;
}

/* ------------------------------------------------------------------ tables needed on the host */

/* per-QP rounding / dead-zone constants of the reference (h264-lab.h:993-1030, 1083-1094) */
static const uint16_t k_rnd_inter[52] = {
    11665, 11665, 11665, 11665, 11665, 11665, 11665, 11665, 11665, 11665, 11665, 12868, 14071, 15273, 16476, 17679, 17740, 17801,
    17863, 17924, 17985, 17445, 16904, 16364, 15823, 15283, 15198, 15113, 15027, 14942, 14857, 15667, 16478, 17288, 18099, 18909,
    19213, 19517, 19822, 20126, 20430, 16344, 12259, 8173, 4088, 4088, 4088, 4088, 4088, 4088, 4088, 4088 };
static const uint16_t k_thr_inter[52] = {
    31878, 31878, 31878, 31878, 31878, 31878, 31878, 31878, 31878, 31878, 31878, 33578, 35278, 36978, 38678, 40378, 41471, 42563,
    43656, 44748, 45841, 46432, 47024, 47615, 48207, 48798, 49354, 49911, 50467, 51024, 51580, 51580, 51580, 51580, 51580, 51580,
    52222, 52864, 53506, 54148, 54790, 45955, 37120, 28286, 19451, 10616, 9326, 8036, 6745, 5455, 4165, 4165 };
static const uint16_t k_thr_inter2[52] = {
    45352, 45352, 45352, 45352, 45352, 45352, 45352, 45352, 45352, 45352, 45352, 41100, 36848, 32597, 28345, 24093, 25904, 27715,
    29525, 31336, 33147, 33429, 33711, 33994, 34276, 34558, 32902, 31246, 29590, 27934, 26278, 26989, 27700, 28412, 29123, 29834,
    29038, 28242, 27445, 26649, 25853, 23440, 21028, 18615, 16203, 13790, 11137, 8484, 5832, 3179, 526, 526 };
static const uint16_t k_deadzonei[52] = {
    3419, 3419, 3419, 3419, 3419, 3419, 3419, 3419, 3419, 3419, 30550, 8845, 14271, 19698, 25124, 30550, 29556, 28562, 27569, 26575,
    25581, 25284, 24988, 24691, 24395, 24098, 24116, 24134, 24153, 24171, 24189, 24010, 23832, 23653, 23475, 23296, 23569, 23842,
    24115, 24388, 24661, 19729, 14797, 9865, 4933, 24661, 3499, 6997, 10495, 13993, 17491, 17491 };
static const uint8_t k_qpc[52] = {
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32, 32,
    33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39 };
/* rate-control model: bits per macroblock at QP 10..50 for P / I frames (h264-lab.h:933-938) */
static const uint16_t k_bits_per_mb[2][41] = {
    { 664, 597, 530, 484, 432, 384, 341, 297, 262, 235, 198, 173, 153, 131, 114, 102, 84, 74, 64, 54, 47, 42, 35, 31, 26, 22, 20, 17,
      15, 13, 12, 10, 9, 9, 7, 7, 6, 5, 4, 1, 1 },
    { 1057, 975, 925, 868, 803, 740, 694, 630, 586, 547, 496, 457, 420, 378, 345, 318, 284, 258, 234, 210, 190, 178, 155, 141, 129,
      115, 102, 95, 82, 75, 69, 60, 55, 51, 45, 41, 40, 35, 31, 28, 24 } };

/* h264-lab.h:5822-5834 */
static uint16_t rnd2thr(int round, int q)
{
    int b, thr = 0;
    for (b = 0x8000; b; b >>= 1)
        if ((thr | b)*q <= 0x10000 - round) thr |= b;
    return (uint16_t)thr;
}

/* h264-lab.h:5839-5912 rc_set_qp: quantizer tables for luma [0] and chroma [1] */
static void build_qdat(uint16_t qdat[2][42], int qp, int p_slice)
{
    static const int16_t qc[6][6] = {
        { 13107, 10, 8066, 13, 5243, 16 }, { 11916, 11, 7490, 14, 4660, 18 }, { 10082, 13, 6554, 16, 4194, 20 },
        {  9362, 14, 5825, 18, 3647, 23 }, {  8192, 16, 5243, 20, 3355, 25 }, {  7282, 18, 4559, 23, 2893, 29 } };
    int c, i, k;
    for (c = 0; c < 2; c++)
    {
        uint16_t *d = qdat[c];
        int div6 = qp*86 >> 9, mod6 = qp - div6*6;
        for (i = 0; i < 3; i++)
        {
            d[2*i]     = (uint16_t)(qc[mod6][2*i] << 1 >> div6);
            d[2*i + 1] = (uint16_t)(qc[mod6][2*i + 1] << div6);
        }
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
}

/* ------------------------------------------------------------------ header bits / NAL */

typedef struct { uint64_t acc; int n; uint8_t *buf; size_t pos; } hbits_t;     /* small MSB-first writer */

static void hb_put(hbits_t *b, int n, uint32_t v)
{
    b->acc = (b->acc << n) | v;
    b->n += n;
    while (b->buf && b->n >= 8)
    {
        b->n -= 8;
        b->buf[b->pos++] = (uint8_t)(b->acc >> b->n);
    }
}
static void hb_ue(hbits_t *b, uint32_t v)
{
    int size = 0;
    uint32_t t = v + 1;
    do size++; while (t >>= 1);
    hb_put(b, 2*size - 1, v + 1);
}
static void hb_se(hbits_t *b, int v) { hb_ue(b, (uint32_t)(v > 0 ? 2*v - 1 : -2*v)); }

/* h264-lab.h:3926-4022: 4-byte start code + payload with emulation prevention; returns bytes written */
static size_t nal_emit(uint8_t *d, const uint8_t *p, size_t n)
{
    size_t i, j = 4;
    int zeros = 0;
    d[0] = d[1] = d[2] = 0; d[3] = 1;
    for (i = 0; i < n; i++)
    {
        if (zeros == 2 && p[i] <= 3) { d[j++] = 3; zeros = 0; }
        zeros = p[i] ? 0 : zeros + 1;
        d[j++] = p[i];
    }
    return j;
}

typedef struct
{
    int width, height, nmbx, nmby, nmb, w, h, cropping;
    int vbv_size_bytes, sps_id;
} seq_t;

static void seq_init(seq_t *s, int width, int height, int vbv, int sps_id)
{
    s->width = width; s->height = height;
    s->nmbx = (width + 15) >> 4; s->nmby = (height + 15) >> 4; s->nmb = s->nmbx*s->nmby;
    s->w = s->nmbx*16; s->h = s->nmby*16;
    s->cropping = !!((width | height) & 15);
    s->vbv_size_bytes = vbv; s->sps_id = sps_id;
}

/* h264-lab.h:4040-4141 encode_sps (profile 66) + h264-lab.h:4147-4176 encode_pps; returns bytes appended to d */
typedef void (*nalu_cb_t)(const unsigned char *nalu_data, int sizeof_nalu_data, void *token);

/* cb (optional): the reference's nal_end() hands EVERY finished NAL -- parameter sets included -- to run_param.nalu_callback,
 * pointer behind the start code, escaped length (h264-lab.h:4014-4018) */
static size_t write_sps_pps(const seq_t *s, int pic_init_qp, uint8_t *d, nalu_cb_t cb, void *token)
{
    static const struct { uint8_t level; uint16_t max_fs, max_vbvdiv5; uint32_t max_dpb; } lim[] = {
        { 10, 99, 175/5, 396 }, { 10, 99, 350/5, 396 }, { 11, 396, 500/5, 900 }, { 12, 396, 1000/5, 2376 },
        { 13, 396, 2000/5, 2376 }, { 20, 396, 2000/5, 2376 }, { 21, 792, 4000/5, 4752 }, { 22, 1620, 4000/5, 8100 },
        { 30, 1620, 10000/5, 8100 }, { 31, 3600, 14000/5, 18000 }, { 32, 5120, 20000/5, 20480 }, { 40, 8192, 25000/5, 32768 },
        { 41, 8192, 62500/5, 32768 }, { 42, 8704, 62500/5, 34816 }, { 50, 22080, 135000/5, 110400 }, { 51, 36864, 240000/5, 184320 } };
    uint8_t tmp[64];
    hbits_t b;
    size_t n = 0;
    int k = 0;
    while (lim[k].level < 51 && (s->nmb > lim[k].max_fs || s->vbv_size_bytes > lim[k].max_vbvdiv5*(5*1000/8) ||
                                 (unsigned)s->nmb > lim[k].max_dpb)) k++;
    memset(&b, 0, sizeof(b)); b.buf = tmp;
    hb_put(&b, 8, 0x67); hb_put(&b, 8, 66); hb_put(&b, 8, 0); hb_put(&b, 8, lim[k].level);
    hb_ue(&b, (uint32_t)s->sps_id);
    hb_ue(&b, 1);                                   /* log2_max_frame_num_minus4 */
    hb_ue(&b, 2);                                   /* pic_order_cnt_type */
    hb_ue(&b, 1);                                   /* num_ref_frames */
    hb_put(&b, 1, 0);
    hb_ue(&b, (uint32_t)(s->nmbx - 1));
    hb_ue(&b, (uint32_t)(s->nmby - 1));
    hb_put(&b, 3, (uint32_t)(6 + s->cropping));
    if (s->cropping)
    {
        hb_ue(&b, 0); hb_ue(&b, (uint32_t)((s->w - s->width) >> 1));
        hb_ue(&b, 0); hb_ue(&b, (uint32_t)((s->h - s->height) >> 1));
    }
    hb_put(&b, 1, 0);
    hb_put(&b, 1, 1);
    if (b.n) hb_put(&b, 8 - b.n, 0);
    n += nal_emit(d + n, tmp, b.pos);
    if (cb) cb(d + 4, (int)(n - 4), token);

    memset(&b, 0, sizeof(b)); b.buf = tmp;
    hb_put(&b, 8, 0x68);
    hb_ue(&b, (uint32_t)(s->sps_id*4)); hb_ue(&b, (uint32_t)s->sps_id);
    hb_put(&b, 1, 0); hb_put(&b, 1, 0);
    hb_ue(&b, 0); hb_ue(&b, 0); hb_ue(&b, 0);
    hb_put(&b, 1, 0); hb_put(&b, 2, 0);
    hb_se(&b, pic_init_qp - 26);
    hb_put(&b, 5, 0x1C);
    hb_put(&b, 1, 1);
    if (b.n) hb_put(&b, 8 - b.n, 0);
    {
        const size_t n0 = n;
        n += nal_emit(d + n, tmp, b.pos);
        if (cb) cb(d + n0 + 4, (int)(n - n0 - 4), token);
    }
    return n;
}

/* h264-lab.h:4182-4333 encode_slice_header for KEY / P frames as a template: the NAL header byte, then (written by the
 * kernel for every slice) ue(first_mb_in_slice), then the tail returned here.  nslices > 1: the row-band build's
 * disable_deblocking_filter_idc = 2 (h264-lab.h:4315-4323) */
static void slice_header_bits(const seq_t *s, int key, int frame_num, int idr_pic_id, int qp, int pic_init_qp, int no_deblock, int nslices,
                              int *nal, uint64_t *bits, int *nbits)
{
    hbits_t b;
    const int idc = nslices > 1 ? (no_deblock ? 1 : 2) : no_deblock;
    memset(&b, 0, sizeof(b));
    *nal = key ? 0x65 : 0x61;
    hb_ue(&b, key ? SLICE_I : SLICE_P);
    hb_ue(&b, (uint32_t)(s->sps_id*4));
    hb_put(&b, 5, (uint32_t)(frame_num & 31));
    if (key) hb_ue(&b, (uint32_t)idr_pic_id);
    if (!key) hb_put(&b, 2, 0);
    if (key) hb_put(&b, 2, 0); else hb_put(&b, 1, 0);
    hb_se(&b, qp - pic_init_qp);
    hb_ue(&b, (uint32_t)idc);
    if (no_deblock != 1) hb_put(&b, 2, 3);
    *bits = b.acc;
    *nbits = b.n;
}

/* the slices of one frame as the kernel exports them (h264e_hip_result_t): complete NALs -- start code + payload with the
 * emulation-prevention bytes already inserted on the device (enc_row.h nal_escape_copy) -- behind each other at 16-byte
 * aligned offsets; they are concatenated in order (h264-lab.h:6565-6569), cb as in nal_end.  Returns bytes written, 0 when
 * cap is too small */
static size_t emit_slices(uint8_t *d, size_t cap, const uint8_t *nals, const h264e_hip_result_t *r, nalu_cb_t cb, void *token)
{
    size_t pos = 0, off = 0;
    int k;
    for (k = 0; k < r->nslices; k++)
    {
        const size_t n = r->slice_nbytes[k];
        if (n < 5 || pos + n > cap) return 0;
        memcpy(d + pos, nals + off, n);
        if (cb) cb(d + pos + 4, (int)(n - 4), token);
        pos += n;
        off += (n + 15) & ~(size_t)15;
    }
    return pos;
}

/* ------------------------------------------------------------------ mv_clusters validation */

static int mvx(int32_t v) { return (int16_t)(v & 0xffff); }
static int mvy(int32_t v) { return (int16_t)((uint32_t)v >> 16); }
static int32_t mvmk(int x, int y) { return (int32_t)(((uint32_t)y << 16) | ((uint32_t)x & 0xffff)); }
static int32_t mvround(int32_t a) { return mvmk((mvx(a) + 1) & ~3, (mvy(a) + 1) & ~3); }

/* h264-lab.h:5263-5278 mv_clusters_update */
static void clusters_step(int32_t c[2], int32_t mv)
{
    int n = mvx(mv)*mvx(mv) + mvy(mv)*mvy(mv);
    int n0 = mvx(c[0])*mvx(c[0]) + mvy(c[0])*mvy(c[0]), n1 = mvx(c[1])*mvx(c[1]) + mvy(c[1])*mvy(c[1]);
    if (n < n1) c[0] = mvmk((63*mvx(c[0]) + mvx(mv) + 32) >> 6, (63*mvy(c[0]) + mvy(mv) + 32) >> 6);
    if (n >= n0) c[1] = mvmk((63*mvx(c[1]) + mvx(mv) + 32) >> 6, (63*mvy(c[1]) + mvy(mv) + 32) >> 6);
}

/*
 * Walk one frame's macroblock records from state `c` (updated in place).  `used` is what the kernel was
 * given: one pair for the whole frame (per_mb = 0) or one pair per macroblock.  traj (optional, [nmb][2])
 * receives the exact value in front of every macroblock.  Returns the first macroblock whose consumed,
 * rounded candidates differ from the exact ones, or -1.
 * nslices > 1 (row bands): every band is encoded by a copy of the encoder that is thrown away (h264-lab.h:6526), so the
 * walk restarts from the frame's start state at every band and `c` comes back unchanged.
 */
static int clusters_walk(int32_t c[2], const h264e_hip_mbrec_t *rec, int nmbx, int nmby, int nslices, const int32_t *used, int per_mb, int32_t *traj)
{
    const int32_t c0[2] = { c[0], c[1] };
    int band, row0 = 0, first_bad = -1;
    if (nslices < 1) nslices = 1;
    for (band = 0; band < nslices; band++)
    {
        const int row1 = row0 + (nmby - row0)/(nslices - band);
        int i;
        c[0] = c0[0]; c[1] = c0[1];
        for (i = row0*nmbx; i < row1*nmbx; i++)
        {
            const int32_t *u = per_mb ? used + 2*i : used;
            if (traj) { traj[2*i] = c[0]; traj[2*i + 1] = c[1]; }
            if (rec[i].used_cand && first_bad < 0 && (mvround(u[0]) != mvround(c[0]) || mvround(u[1]) != mvround(c[1]))) first_bad = i;
            if (rec[i].type < 5) clusters_step(c, rec[i].mv0);
        }
        row0 = row1;
    }
    if (nslices > 1) { c[0] = c0[0]; c[1] = c0[1]; }
    return first_bad;
}

/* ------------------------------------------------------------------ rate control (frame level) */

typedef struct { int qp, prev_qp, vbv_bits, qp_smooth, dqp_smooth, max_dqp, bit_budget, vbv_target_level; } rc_t;

static uint32_t mul32x32shr16(uint32_t x, uint32_t y)               /* h264-lab.h:3420 */
{
    return (x >> 16)*(y & 0xFFFFu) + x*(y >> 16) + ((y & 0xFFFFu)*(x & 0xFFFFu) >> 16);
}
static uint32_t div_q16(uint32_t numer, uint32_t denum)             /* h264-lab.h:3430 */
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

/* h264-lab.h:5924-6070 rc_frame_start without long-term references; returns the frame QP */
static int rc_frame_start(rc_t *rc, int gop, int nmb, int vbv_size_bytes, int desired_frame_bytes, int qp_min, int qp_max, int is_intra)
{
    unsigned np = (unsigned)(gop - 1u) < 63u ? (unsigned)(gop - 1u) : 63u;
    int qp = -1, add_bits, bit_budget = desired_frame_bytes*8, nominal_p, gop_bits, stationary;
    uint32_t peak_q16;
    do
    {
        qp++;
        gop_bits = (int)(k_bits_per_mb[0][qp]*np + k_bits_per_mb[1][qp]);
    } while (gop_bits*nmb > (int)(np + 1)*desired_frame_bytes*8 && qp < 40);
    peak_q16 = div_q16((uint32_t)k_bits_per_mb[1][qp] << 16, (uint32_t)k_bits_per_mb[0][qp] << 16);
    if (np)
    {
        uint32_t ratio = div_q16((np + 1) << 16, (np << 16) + peak_q16);
        nominal_p = (int)mul32x32shr16((uint32_t)(desired_frame_bytes*8), ratio);
    } else
        nominal_p = 0;
    stationary = imin(vbv_size_bytes*8 >> 4, desired_frame_bytes*8);
    if (is_intra)
        add_bits = (int)mul32x32shr16((uint32_t)nominal_p, peak_q16) - bit_budget;
    else
    {
        add_bits = nominal_p - bit_budget;
        if (vbv_size_bytes) add_bits += (rc->vbv_target_level - rc->vbv_bits) >> 4;
    }
    if (vbv_size_bytes) add_bits = imin(add_bits, (vbv_size_bytes*8*7 >> 3) - rc->vbv_bits);
    bit_budget += add_bits;
    bit_budget = imin(bit_budget, desired_frame_bytes*8*16);
    bit_budget = imax(bit_budget, desired_frame_bytes*8 >> 2);
    if (is_intra) rc->vbv_target_level = rc->vbv_bits + bit_budget - desired_frame_bytes*8;
    rc->vbv_target_level -= desired_frame_bytes*8 - nominal_p;
    rc->vbv_target_level = imax(rc->vbv_target_level, stationary);
    rc->bit_budget = bit_budget;
    {
        const uint16_t *bits = k_bits_per_mb[!!is_intra];
        for (qp = 0; qp < 42 - 1; qp++)
            if (bits[qp]*nmb < bit_budget) break;
    }
    qp += 10;
    qp += rc->dqp_smooth;
    if (rc->prev_qp > qp + 1) qp = (rc->prev_qp + qp + 1)/2;
    qp = imin(qp, qp_max); qp = imax(qp, qp_min); qp = imin(qp, 51);             /* h264-lab.h:5841-5843 */
    rc->qp = qp;
    rc->qp_smooth = qp << 8;
    rc->prev_qp = qp;
    return qp;
}

/* h264-lab.h:6075-6141 rc_frame_end (stuffing / empty-frame options are refused at init) */
static void rc_frame_end(rc_t *rc, int nmb, int vbv_size_bytes, int desired_frame_bytes, int out_bytes, int intra, int all_skipped)
{
    if (!all_skipped)
    {
        int qp;
        for (qp = 0; qp != 41 && k_bits_per_mb[intra][qp]*nmb > out_bytes*8 - 32; qp++) {}
        qp += 10;
        if ((rc->qp_smooth >> 8) - rc->dqp_smooth < qp - 1) rc->dqp_smooth--;
        else if ((rc->qp_smooth >> 8) - rc->dqp_smooth > qp + 1) rc->dqp_smooth++;
        if (intra) rc->max_dqp = rc->dqp_smooth;
        else rc->max_dqp = imax(rc->max_dqp, (rc->qp_smooth >> 8) - qp);
    }
    rc->vbv_bits += out_bytes*8 - desired_frame_bytes*8;
    if (vbv_size_bytes)
    {
        if (rc->vbv_bits < 0) rc->vbv_bits = 0;
        if (rc->vbv_bits > vbv_size_bytes*8) rc->vbv_bits = vbv_size_bytes*8;
    } else
        rc->vbv_bits = 0;
}

/* ------------------------------------------------------------------ one frame on one chain, validated */

/*
 * One step (one frame of every active chain) with the mv_clusters speculation made exact.
 * tasks[k].mv_clusters holds the state in front of chain k's frame (run[k]); the kernel uses it for every
 * macroblock.  Chains whose frame moves the state are walked exactly; where a consumed rounded candidate
 * differs, those chains (all of them together, one launch) are encoded again with the walked per-macroblock
 * values, until the walk is consistent -- every pass fixes at least the first offending macroblock of each
 * chain, so this terminates.  run[k] is advanced to the state behind the frame; arr_out[k] (optional) receives
 * the per-macroblock array the final pass used (caller frees) or NULL.
 */
static int step_exact(h264e_hip_pool_t *pool, int nchains, h264e_hip_task_t *tasks, int nmbx, int nmby, int32_t (*run)[2],
                      int32_t **arr_out, int *extra_passes)
{
    const int nmb = nmbx*nmby;
    int *flags = (int *)calloc(2*(size_t)nchains, sizeof(int));
    int32_t **arr = (int32_t **)calloc((size_t)nchains, sizeof(int32_t *));
    char *todo = (char *)calloc((size_t)nchains, 1);
    h264e_hip_mbrec_t *rec = NULL;
    int32_t *traj = NULL;
    int rc = -1, k, pass, pending = 0;
    if (!flags || !arr || !todo) goto done;
    for (k = 0; k < nchains; k++)
    {
        tasks[k].mv_clusters_per_mb = NULL;
        if (tasks[k].active) { tasks[k].mv_clusters[0] = run[k][0]; tasks[k].mv_clusters[1] = run[k][1]; todo[k] = 1; pending++; }
    }
    for (pass = 0; pending; pass++)
    {
        if (pass > nmb + 2) { snprintf(g_host_err, sizeof(g_host_err), "mv_clusters re-encode does not converge"); goto done; }
        if (h264e_hip_submit(pool, tasks) || h264e_hip_sync(pool)) goto done;
        for (k = 0; k < nchains; k++)
        {
            int32_t cc[2];
            int bad;
            h264e_hip_result_t r1;
            const h264e_hip_mbrec_t *recs;
            if (!todo[k]) continue;
            /* results come through host-mapped memory, written by the frame's finalizer workgroup (no device-to-host copies) */
            if (h264e_hip_stream_done(pool, k, &r1) != 1) { snprintf(g_host_err, sizeof(g_host_err), "frame did not complete"); goto done; }
            flags[2*k] = r1.clusters_moved; flags[2*k + 1] = r1.overflow;
            if (flags[2*k + 1]) { snprintf(g_host_err, sizeof(g_host_err), "bit buffer overflow"); goto done; }
            if (!arr[k] && !flags[2*k]) { todo[k] = 0; tasks[k].active = 0; pending--; continue; }   /* fixed point: state unchanged */
            if (!traj)
            {
                traj = (int32_t *)malloc(sizeof(int32_t)*2*(size_t)nmb);
                if (!traj) goto done;
            }
            recs = h264e_hip_stream_mbrec(pool, k);
            if (!recs) goto done;
            cc[0] = run[k][0]; cc[1] = run[k][1];
            bad = clusters_walk(cc, recs, nmbx, nmby, tasks[k].nslices, arr[k] ? arr[k] : run[k], arr[k] != NULL, traj) >= 0;
            if (!bad)
            {
                run[k][0] = cc[0]; run[k][1] = cc[1];
                todo[k] = 0; tasks[k].active = 0; pending--;
                continue;
            }
            if (!arr[k]) arr[k] = (int32_t *)malloc(sizeof(int32_t)*2*(size_t)nmb);
            if (!arr[k]) goto done;
            memcpy(arr[k], traj, sizeof(int32_t)*2*(size_t)nmb);
            tasks[k].mv_clusters_per_mb = arr[k];
            if (h264e_hip_rewind_frame(pool, k, tasks[k].frame_slot)) goto done;
        }
    }
    if (extra_passes) *extra_passes = pass > 0 ? pass - 1 : 0;
    rc = 0;
done:
    if (rc) h264e_hip_release(pool);
    for (k = 0; k < nchains && arr; k++)
    {
        if (!rc && arr_out) arr_out[k] = arr[k]; else free(arr[k]);
    }
    free(flags); free(arr); free(todo); free(rec); free(traj);
    return rc;
}

/* ------------------------------------------------------------------ drop-in API */

typedef struct
{
    uint32_t magic, serial;
    int impl;                               /* index into g_impl: device resources live OUTSIDE the caller's blob */
    H264E_create_param_t param;
    H264E_run_param_t run_param;
    seq_t seq;
    int frame_num, next_idr_pic_id, pic_init_qp;
    int32_t clusters[2];
    rc_t rc;
    int slices;                             /* row-band slices per frame (H264E_set_slices), 0 / 1 = one */
} henc_t;

/* The reference API has no destructor and callers simply free() the blob, so nothing that needs
 * releasing may be reachable only through it: pools and staging buffers sit in this registry, keyed by blob address.
 * The registry grows on demand and is guarded by one mutex (H264E_init / H264E_close / H264E_encode's lookup may run on
 * different threads for different encoders, like the reference's re-entrant instances).  An entry is reclaimed when
 * H264E_close is called, when the same blob is initialised again, or when a NEW blob overlaps its address range (the
 * caller freed the old blob and the allocator handed the memory out again: the old encoder cannot be alive). */
typedef struct { void *owner; size_t owner_bytes; uint32_t serial; h264e_hip_pool_t *pool; uint8_t *rbsp; size_t rbsp_cap; uint8_t *recon; } impl_t;
static impl_t *g_impl;
static int g_impl_cap;
static uint32_t g_serial;
static int g_atexit;
static pthread_mutex_t g_reg_lock = PTHREAD_MUTEX_INITIALIZER;

static void impl_release(impl_t *m)
{
    if (m->pool) h264e_hip_pool_destroy(m->pool);
    free(m->rbsp); free(m->recon);
    memset(m, 0, sizeof(*m));
}

static void release_all(void)
{
    int i;
    pthread_mutex_lock(&g_reg_lock);
    for (i = 0; i < g_impl_cap; i++) if (g_impl[i].owner) impl_release(g_impl + i);
    pthread_mutex_unlock(&g_reg_lock);
}

/* copy of the entry (the table may be reallocated by another thread's H264E_init): pointers inside stay valid as long as
 * this encoder is not closed concurrently, which the API forbids per instance as the reference does */
static int impl_of(const henc_t *e, impl_t *out)
{
    int ok = 0;
    if (!e || e->magic != MAGIC) return 0;
    pthread_mutex_lock(&g_reg_lock);
    if (e->impl >= 0 && e->impl < g_impl_cap && g_impl[e->impl].owner == (const void *)e && g_impl[e->impl].serial == e->serial)
    {
        *out = g_impl[e->impl];
        ok = 1;
    }
    pthread_mutex_unlock(&g_reg_lock);
    return ok;
}

/* with the lock held: release every entry whose persist blob [owner, owner + owner_bytes) OVERLAPS [p, p + n): the memory of an abandoned
 * encoder may begin below a new allocation and still reach into it */
static void reclaim_range(const void *p, size_t n)
{
    const char *lo = (const char *)p, *hi = lo + (n ? n : 1);
    int i;
    for (i = 0; i < g_impl_cap; i++)
        if (g_impl[i].owner)
        {
            const char *olo = (const char *)g_impl[i].owner, *ohi = olo + (g_impl[i].owner_bytes ? g_impl[i].owner_bytes : 1);
            if (olo < hi && lo < ohi) impl_release(g_impl + i);
        }
}

void H264E_close(H264E_persist_t *p)
{
    if (!p) return;
    pthread_mutex_lock(&g_reg_lock);
    reclaim_range(p, 1);
    pthread_mutex_unlock(&g_reg_lock);
}

/* h264-lab.h:6252-6286 enc_check_create_params (+ the options this implementation refuses) */
static int check_params(const H264E_create_param_t *par)
{
    if (!par) return H264E_STATUS_BAD_ARGUMENT;
    if ((int)(par->vbv_size_bytes | par->gop) < 0) return H264E_STATUS_BAD_PARAMETER;
    if (par->width <= 0 || par->height <= 0) return H264E_STATUS_BAD_PARAMETER;
    if ((unsigned)(par->const_input_flag | par->fine_rate_control_flag | par->vbv_overflow_empty_frame_flag | par->vbv_underflow_stuffing_flag) > 1)
        return H264E_STATUS_BAD_PARAMETER;
    if ((unsigned)par->max_long_term_reference_frames > 8) return H264E_STATUS_BAD_PARAMETER;
    if ((par->width | par->height) & 1) return H264E_STATUS_SIZE_NOT_MULTIPLE_2;
    if (((par->width | par->height) & 15) && !par->const_input_flag) return H264E_STATUS_SIZE_NOT_MULTIPLE_16;
    return H264E_STATUS_SUCCESS;
}

static int unsupported(const H264E_create_param_t *par)
{
    return par->max_long_term_reference_frames || par->temporal_denoise_flag || par->fine_rate_control_flag ||
           par->vbv_overflow_empty_frame_flag || par->vbv_underflow_stuffing_flag || par->num_layers > 1;
}

/* the reference's blob sizes (h264-lab.h:6191-6230 enc_alloc / enc_alloc_scratch with sizeof(h264e_enc_t) = 1184,
 * sizeof(scratch_t) = 2962 on LP64): encode_app prints them, and callers size their buffers with them */
static void ref_sizes(const H264E_create_param_t *par, int *persist, int *scratch)
{
    int nmbx = (par->width + 15) >> 4, nmby = (par->height + 15) >> 4, p = 1;
    int nref = 1 + par->max_long_term_reference_frames + par->const_input_flag + !!par->temporal_denoise_flag;
    static const int a16 = 15;
    *persist = (((16 + (nmbx + 2)*(nmby + 2)*384*nref) - 1 + 15) & ~15) + 1184;
#define ALLOC(sz) p = (p + a16) & ~a16; p += (sz)
    ALLOC(2962);
    ALLOC(nmbx*nmby*(384 + 2 + 10)*3/2);
    ALLOC(nmbx*8 + 8);
    ALLOC((nmbx*4 + 8)*4);
    ALLOC(nmbx*4 + 4);
    ALLOC(nmbx);
    ALLOC(nmbx);
    ALLOC(nmbx);
    ALLOC(nmbx*32 + 32 + 16);
#undef ALLOC
    *scratch = p - 1;
}

int H264E_sizeof(const H264E_create_param_t *par, int *sizeof_persist, int *sizeof_scratch)
{
    int err = check_params(par);
    if (!sizeof_persist || !sizeof_scratch) err = H264E_STATUS_BAD_ARGUMENT;
    if (err) return err;
    ref_sizes(par, sizeof_persist, sizeof_scratch);
    return H264E_STATUS_SUCCESS;
}

int H264E_init(H264E_persist_t *p, const H264E_create_param_t *par)
{
    henc_t *e = (henc_t *)p;
    impl_t fresh;
    int i, sp, ss, err = check_params(par);
    g_host_err[0] = 0;
    if (!e) return H264E_STATUS_BAD_ARGUMENT;
    if (err) return err;
    if (unsupported(par))
    {
        snprintf(g_host_err, sizeof(g_host_err), "option outside the MI355X encode path (long-term refs, denoise, MB-level RC, VBV stuffing/empty frames, SVC)");
        return H264E_STATUS_BAD_PARAMETER;
    }
    ref_sizes(par, &sp, &ss);
    /* whatever lived in this memory goes FIRST (re-init of the same blob, or of memory an abandoned encoder lived in): a re-init does not
     * need the device memory twice, and a failure below does not leave an entry registered for a blob that has been zeroed */
    pthread_mutex_lock(&g_reg_lock);
    reclaim_range(p, (size_t)sp);
    pthread_mutex_unlock(&g_reg_lock);
    memset(e, 0, sizeof(*e));
    e->param = *par;
    seq_init(&e->seq, par->width, par->height, par->vbv_size_bytes, par->sps_id);
    /* device resources are created outside the lock (slow), registered under it */
    memset(&fresh, 0, sizeof(fresh));
    if (h264e_hip_pool_create(&fresh.pool, pick_device(), par->width, par->height, 1, 1, 1))
        return H264E_STATUS_BAD_ARGUMENT;       /* no device: the HIP path is the only path */
    fresh.rbsp_cap = (size_t)e->seq.nmb*660 + 8192;
    fresh.rbsp = (uint8_t *)malloc(fresh.rbsp_cap);
    if (!par->const_input_flag) fresh.recon = (uint8_t *)malloc((size_t)e->seq.w*e->seq.h*3/2);
    fresh.owner = e; fresh.owner_bytes = (size_t)sp;
    pthread_mutex_lock(&g_reg_lock);
    reclaim_range(p, (size_t)sp);               /* (another thread may have registered an overlapping blob meanwhile) */
    for (i = 0; i < g_impl_cap && g_impl[i].owner; i++) {}
    if (i == g_impl_cap)
    {
        const int ncap = g_impl_cap ? 2*g_impl_cap : 16;
        impl_t *t = (impl_t *)realloc(g_impl, sizeof(impl_t)*(size_t)ncap);
        if (!t)
        {
            pthread_mutex_unlock(&g_reg_lock);
            impl_release(&fresh);
            snprintf(g_host_err, sizeof(g_host_err), "out of host memory");
            return H264E_STATUS_BAD_ARGUMENT;
        }
        memset(t + g_impl_cap, 0, sizeof(impl_t)*(size_t)(ncap - g_impl_cap));
        g_impl = t; g_impl_cap = ncap;
    }
    fresh.serial = e->serial = ++g_serial;
    g_impl[i] = fresh;
    e->impl = i;
    e->magic = MAGIC;
    if (!g_atexit) { atexit(release_all); g_atexit = 1; }
    pthread_mutex_unlock(&g_reg_lock);
    return H264E_STATUS_SUCCESS;
}

int H264E_set_slices(H264E_persist_t *p, int nslices)
{
    henc_t *e = (henc_t *)p;
    if (!e || e->magic != MAGIC || nslices < 0 || nslices > H264E_HIP_MAX_SLICES) return H264E_STATUS_BAD_PARAMETER;
    e->slices = nslices;
    return H264E_STATUS_SUCCESS;
}

void H264E_set_vbv_state(H264E_persist_t *p, int vbv_size_bytes, int vbv_fullness_bytes)
{
    henc_t *e = (henc_t *)p;
    if (!e || e->magic != MAGIC) return;
    e->param.vbv_size_bytes = vbv_size_bytes;
    e->seq.vbv_size_bytes = vbv_size_bytes;
    if (vbv_fullness_bytes >= 0)
    {
        e->rc.vbv_bits = vbv_fullness_bytes*8;
        e->rc.vbv_target_level = e->rc.vbv_bits;
    }
}

int H264E_encode(H264E_persist_t *p, H264E_scratch_t *scratch, const H264E_run_param_t *opt, H264E_io_yuv_t *in,
                 unsigned char **coded_data, int *sizeof_coded_data)
{
    henc_t *e = (henc_t *)p;
    impl_t mm, *m = impl_of(e, &mm) ? &mm : NULL;
    uint8_t *out = (uint8_t *)scratch;
    size_t out_pos = 0, cap;
    int frame_type, key, qp, sp, ss;
    h264e_hip_task_t task;
    h264e_hip_result_t res;
    const uint8_t *yuv[3];
    g_host_err[0] = 0;
    if (!m || !scratch || !in || !coded_data || !sizeof_coded_data) return H264E_STATUS_BAD_ARGUMENT;
    ref_sizes(&e->param, &sp, &ss);
    cap = (size_t)ss;
    if (opt) e->run_param = *opt;
    opt = &e->run_param;
    if (!e->run_param.qp_max || e->run_param.qp_max > 51) e->run_param.qp_max = 51;       /* h264-lab.h:6707-6715 */
    if (!e->run_param.qp_min || e->run_param.qp_min < 10) e->run_param.qp_min = 10;
    if (opt->desired_nalu_bytes)
    {
        snprintf(g_host_err, sizeof(g_host_err), "desired_nalu_bytes (multi-slice) is outside the MI355X encode path");
        return H264E_STATUS_BAD_ARGUMENT;
    }

    frame_type = opt->frame_type;
    if (frame_type == H264E_FRAME_TYPE_DEFAULT) frame_type = e->frame_num ? H264E_FRAME_TYPE_P : H264E_FRAME_TYPE_KEY;
    if (frame_type != H264E_FRAME_TYPE_KEY && frame_type != H264E_FRAME_TYPE_P) return H264E_STATUS_BAD_FRAME_TYPE;
    key = frame_type == H264E_FRAME_TYPE_KEY;
    if (key)
    {
        e->pic_init_qp = imax(imin(30, e->run_param.qp_max), e->run_param.qp_min);       /* h264-lab.h:6768-6775 */
        e->next_idr_pic_id ^= 1;
        e->frame_num = 0;
        out_pos += write_sps_pps(&e->seq, e->pic_init_qp, out + out_pos, e->run_param.nalu_callback, e->run_param.nalu_callback_token);
    } else if (!e->pic_init_qp)
        return H264E_STATUS_BAD_FRAME_TYPE;                                              /* h264-lab.h:6801-6804 */

    qp = rc_frame_start(&e->rc, e->param.gop, e->seq.nmb, e->param.vbv_size_bytes, opt->desired_frame_bytes,
                        e->run_param.qp_min, e->run_param.qp_max, key);

    memset(&task, 0, sizeof(task));
    task.active = 1; task.frame_index = 0; task.frame_slot = 0;
    task.slice_type = key ? SLICE_I : SLICE_P;
    task.qp = qp; task.speed = opt->encode_speed;
    task.nslices = e->slices > 1 ? imin(e->slices, e->seq.nmby) : 1;
    slice_header_bits(&e->seq, key, e->frame_num, e->next_idr_pic_id, qp, e->pic_init_qp,
                      (opt->encode_speed == 8 || opt->encode_speed == 10), task.nslices, &task.hdr_nal, &task.hdr_bits, &task.hdr_nbits);
    build_qdat(task.qdat, qp, !key);

    yuv[0] = in->yuv[0]; yuv[1] = in->yuv[1]; yuv[2] = in->yuv[2];
    memset(&res, 0, sizeof(res));
    if (e->param.vbv_size_bytes && e->rc.vbv_bits - opt->desired_frame_bytes*8 > e->param.vbv_size_bytes*8)
    {
        /* h264-lab.h:6497-6510 "encode transparent frame on VBV overflow" -- reachable only right after H264E_set_vbv_state (rc_frame_end
         * clamps the fullness to the VBV size): one slice whose whole payload is a skip run over the picture (written for key frames
         * too), the reference picture as the reconstruction.  Nothing for the device to do: the pool's reference / reconstruction pair is
         * simply not swapped, so the next frame predicts from the same picture; mv_clusters do not move (no macroblock was encoded). */
        uint8_t rb[32];
        hbits_t b;
        size_t w;
        memset(&b, 0, sizeof(b));
        b.buf = rb;
        hb_put(&b, 8, (uint32_t)task.hdr_nal);
        hb_ue(&b, 0);                                                   /* first_mb_in_slice */
        hb_put(&b, task.hdr_nbits > 32 ? task.hdr_nbits - 32 : 0, (uint32_t)(task.hdr_nbits > 32 ? task.hdr_bits >> 32 : 0));
        hb_put(&b, task.hdr_nbits > 32 ? 32 : task.hdr_nbits, (uint32_t)task.hdr_bits);
        hb_ue(&b, (uint32_t)e->seq.nmb);                                /* mb_skip_run = every macroblock */
        hb_put(&b, 1, 1);                                               /* rbsp_stop_one_bit */
        if (b.n) hb_put(&b, 8 - b.n, 0);
        if (out_pos + 2*b.pos + 8 > cap) { snprintf(g_host_err, sizeof(g_host_err), "coded frame does not fit the scratch blob"); return H264E_STATUS_BAD_ARGUMENT; }
        w = nal_emit(out + out_pos, rb, b.pos);
        if (opt->nalu_callback) opt->nalu_callback(out + out_pos + 4, (int)(w - 4), opt->nalu_callback_token);
        out_pos += w;
        res.all_skipped = 1;
    } else
    {
    if (h264e_hip_reset_results(m->pool, 0) || h264e_hip_upload_planes(m->pool, 0, yuv, in->stride)) return H264E_STATUS_BAD_ARGUMENT;
    {
        int32_t run[1][2] = { { e->clusters[0], e->clusters[1] } };
        if (step_exact(m->pool, 1, &task, e->seq.nmbx, e->seq.nmby, run, NULL, NULL) || h264e_hip_stream_done(m->pool, 0, &res) != 1) return H264E_STATUS_BAD_ARGUMENT;
        e->clusters[0] = run[0][0]; e->clusters[1] = run[0][1];
    }
    {
        const uint8_t *nals = h264e_hip_stream_rbsp(m->pool, 0);
        size_t w;
        if (res.in_device)
        {
            if (res.nbytes > m->rbsp_cap || h264e_hip_stream_fetch_nals(m->pool, 0, m->rbsp, res.nbytes)) return H264E_STATUS_BAD_ARGUMENT;
            nals = m->rbsp;
        }
        w = emit_slices(out + out_pos, cap - out_pos, nals, &res, opt->nalu_callback, opt->nalu_callback_token);
        if (!w)
        {
            snprintf(g_host_err, sizeof(g_host_err), "coded frame does not fit the scratch blob");
            return H264E_STATUS_BAD_ARGUMENT;
        }
        out_pos += w;
    }
    }

    rc_frame_end(&e->rc, e->seq.nmb, e->param.vbv_size_bytes, opt->desired_frame_bytes, (int)out_pos, key, res.all_skipped);

    if (!e->param.const_input_flag)
    {
        /* h264-lab.h:6719-6723: the reconstruction replaces the caller's input picture (encode_app --psnr relies on it) */
        int c, y;
        const uint8_t *s = m->recon;
        if (h264e_hip_read_recon(m->pool, 0, m->recon)) return H264E_STATUS_BAD_ARGUMENT;
        for (c = 0; c < 3; c++)
        {
            int w = e->seq.w >> (c ? 1 : 0), h = e->seq.h >> (c ? 1 : 0);
            for (y = 0; y < h; y++) memcpy(in->yuv[c] + (size_t)y*in->stride[c], s + (size_t)y*w, (size_t)w);
            s += (size_t)w*h;
        }
    }
    if (++e->frame_num >= e->param.gop && e->param.gop && e->run_param.frame_type == H264E_FRAME_TYPE_DEFAULT) e->frame_num = 0;
    *coded_data = out;
    *sizeof_coded_data = (int)out_pos;
    return H264E_STATUS_SUCCESS;
}

/* ------------------------------------------------------------------ whole-clip encode: temporal wavefront */

/*
 * The clip is ONE stream, encoded in stream order, but many consecutive frames are in flight at once: each frame is a
 * job of the same launch and starts a few macroblock rows behind the frame it references (h264e_kernels.hip), key
 * frames start immediately.  All frames of a launch speculate the mv_clusters state known when the launch starts;
 * afterwards the host validates them in order (exact walk of the per-macroblock records) and relaunches from the
 * first frame whose consumed candidates differ -- that frame with exact per-macroblock values.
 *
 * The encoder is resumable: H264E_clip_encode() encodes the frames that have been uploaded and not yet encoded (or as many
 * as fit the caller's output buffer) and keeps every piece of stream state -- position, mv_clusters, idr parity, rate
 * control, reference pictures -- for the next call.  Input frames live in a ring of `resident` HBM slots, so a file of any
 * length streams through a bounded amount of host and device memory (encode_app --clip).
 */
struct H264E_clip_tag
{
    H264E_clip_param_t par;
    seq_t seq;
    int nframes, gop_len, ring;
    int resident;                           /* input frames kept in HBM (ring): frame f lives in slot f % resident */
    h264e_hip_pool_t *pool;
    /* ---- stream state, kept across H264E_clip_encode calls */
    int next;                               /* next frame to encode */
    int avail;                              /* frames [0, avail) have been uploaded / generated */
    int pending_avail;                      /* ... once the asynchronous uploads in flight have landed */
    void (*idle_hook)(void *token); void *idle_token;   /* called while the encoder waits for the GPU (the app feeds uploads from it) */
    int32_t state[2];                       /* exact mv_clusters in front of frame `next` */
    int first_dev;                          /* the next launch's first frame is encoded again with the per-macroblock trajectory its walk left on the device */
    int32_t *first_arr;                     /* (unused by the streaming path since the walk moved to the device; kept NULL) */
    int first_row;                          /* ... which restarts at this macroblock row */
    int32_t after[2]; int have_after;       /* predicted state behind that frame */
    int narrow;                             /* reference-window geometry in use (h264e_dev.h) */
    int narrow_ok;                          /* the picture size allows the narrow geometry at all */
    int wide_until, wide_hold;              /* wide geometry until this frame; length of the next wide spell (doubles when narrow fails again) */
    int launch_base, launch_frames;         /* frames per launch: the pipeline depth after a mis-speculation, growing after clean launches (H264E_clip_open) */
    int stopped_before;                     /* the previous launch was stopped before its last frame (statistics: the next one pays a pipeline refill) */
    long long far_acc; int far_frames;      /* far reads / frames since the last decision */
    rc_t rcs; int rc_frame, rc_qp;          /* rate control: state, the frame rc_frame_start has run for, its QP */
    int rc_last_bytes[2];                   /* size of the last accepted P / key frame: what a frame still in flight is predicted to weigh */
    /* keep_records: what every accepted frame consumed, so that a different start state can be validated later (GOP shards) */
    h264e_hip_mbrec_t **rec_store;          /* [nframes] macroblock records of the accepted encode, or NULL */
    int32_t (*used_store)[2];               /* [nframes] frame-constant candidates it was given ... */
    int32_t **permb_store;                  /* [nframes] ... or the per-macroblock trajectory it was given */
    uint64_t *ssd_out;                      /* optional: [3] sums of squared differences input vs reconstruction per encoded frame of a call */
    int recon_floor;                        /* rate control: frames below this one may have been overwritten by hedge leaves (H264E_clip_read_recon) */
    int32_t *traj;                          /* scratch: walked trajectory [nmb][2] */
    uint8_t *big; size_t big_cap;           /* scratch: NALs of a frame that did not fit the host mirror */
    h264e_hip_task_t *tasks;                /* scratch [ring] */
    int32_t (*used)[2];                     /* scratch [ring] */
};

int H264E_struct_size(int which) { return which == 0 ? (int)sizeof(H264E_clip_param_t) : which == 1 ? (int)sizeof(H264E_clip_stats_t) : -1; }

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec*1e3 + ts.tv_nsec*1e-6;
}

void H264E_clip_rewind(H264E_clip_t *c)
{
    if (!c) return;
    c->next = 0;
    c->state[0] = c->par.mv_clusters_in[0]; c->state[1] = c->par.mv_clusters_in[1];
    free(c->first_arr); c->first_arr = NULL; c->first_dev = 0;
    c->first_row = 0; c->have_after = 0; c->recon_floor = 0; c->stopped_before = 0;
    /* reference-window geometry (h264e_dev.h): narrow = consecutive frames 4 macroblock steps apart, as long as vectors rarely
     * reach more than 12 samples right / down of their macroblock; wide (7 steps) for the rest of the clip otherwise.  Large
     * pictures already fill the GPU's resident workgroups with the wide geometry: the narrow one only pays below ~12k macroblocks */
    c->narrow_ok = c->narrow = (getenv("H264E_WIDE_WINDOW") || c->seq.nmb > 12000) ? 0 : 1;
    c->wide_until = 0; c->wide_hold = 30; c->far_acc = 0; c->far_frames = 0;
    /* optimistic where mis-speculations are rare by construction (row bands restart the state, intra frames do not read it): as long
     * as memory allows; else the pipeline depth, growing with every clean launch */
    c->launch_frames = (c->par.slices > 1 || c->gop_len == 1) ? c->ring - 1 : c->launch_base;
    memset(&c->rcs, 0, sizeof(c->rcs));
    c->rc_frame = -1; c->rc_qp = c->par.qp; c->rc_last_bytes[0] = c->rc_last_bytes[1] = 0;
}

int H264E_clip_open(H264E_clip_t **out, const H264E_clip_param_t *par, int nframes)
{
    H264E_clip_t *c;
    g_host_err[0] = 0;
    if (!out || !par || nframes <= 0 || par->width <= 0 || par->height <= 0 || ((par->width | par->height) & 1) || par->gop < 0) return -1;
    c = (H264E_clip_t *)calloc(1, sizeof(*c));
    if (!c) return -1;
    c->par = *par;
    c->par.qp = imin(imax(par->qp, 10), 51);
    seq_init(&c->seq, par->width, par->height, par->vbv_size_bytes, 0);
    c->nframes = nframes;
    c->gop_len = par->gop ? par->gop : (1 << 30);
    /* ring of picture / result slots = frames per launch + 1.  Not bounded by residency: workgroups only wait for lower
     * block indices, so a launch larger than the GPU simply streams through it in order. */
    /* default: 96 frames per launch at 1080p and above; small pictures have short pipelines and cheap slots, so they get
     * proportionally more (fewer launch boundaries), up to 1024 */
    {
        const int nmb = c->seq.nmb;
        /* large pictures: the ~2000 resident workgroups hold only a few frames (4K 33, 8K 16 at one workgroup per row), so slots
         * beyond that only cost memory: at most 6500 rows' worth */
        /* (row-band slices: relaunches are rare and every frame offers several wavefronts, twice as many slots pay) */
        /* (measured: 1080p 160 frames per launch after an event beat 96 by 2.5 %, 4K 96 beat 49 by 3 %; 8K is bound by the resident
         * workgroups, 25 frames beat 49 by 8 %) */
        const int rows_budget = (par->slices > 1 || c->seq.nmby <= 135) ? 13000 : 6500;
        c->launch_base = par->max_chains > 0 ? par->max_chains : imax(imin(161, imax(13, rows_budget/imax(c->seq.nmby, 1) + 1)), imin(1025, 97*8160/imax(nmb, 1))) - 1;
        /* That is the pipeline's depth: what a launch needs to keep the chip busy, and what a launch gets right after a
         * mis-speculation (everything behind the failed frame is thrown away, so a short launch wastes less set-up and drain).
         * But a launch BOUNDARY without an event costs a pipeline drain and refill too (one frame latency), so event-free stretches
         * -- row-band slices, all-intra streams, quiet content -- want launches as long as memory allows: the ring is sized by a
         * memory budget (896 MB of host-mapped mirrors, 24 GB of HBM) and the frames per launch grow again after every clean launch
         * (measured at 1080p x 600: 8 slices 14.5 -> 18.3 M MB/s, all-intra 14.2 -> 19.6 M, single slice 6.77 -> 6.94 M). */
        {
            const double host_slot = (double)nmb*168.0 + 65536.0 + 4096.0;
            const double dev_slot = (double)nmb*256.0*3.0 + (double)nmb*(64 + 96 + 8 + 640 + 660 + 2048 + 64) + 65536.0;      /* two pictures, records, row bit buffers (2 KB per macroblock), arenas */
            const double by_host = 896.0*1048576.0/host_slot, by_dev = 24.0*1073741824.0/dev_slot;
            int cap = (int)(by_host < by_dev ? by_host : by_dev);
            cap = imin(imax(cap, c->launch_base), 1024);
            c->ring = (par->max_chains > 0 ? par->max_chains : cap) + 1;
        }
    }
    if (getenv("H264E_RING")) { c->ring = atoi(getenv("H264E_RING")); c->launch_base = imin(c->launch_base, imax(c->ring - 1, 1)); }      /* experiments */
    c->ring = imax(2, imin(c->ring, nframes + 1));
    if (getenv("H264E_LAUNCH_BASE")) c->launch_base = atoi(getenv("H264E_LAUNCH_BASE"));      /* experiments */
    c->launch_base = imax(1, imin(c->launch_base, c->ring - 1));
    c->resident = par->resident_frames > 0 ? imin(par->resident_frames, nframes) : nframes;
    c->traj = (int32_t *)malloc(sizeof(int32_t)*2*(size_t)c->seq.nmb);
    c->tasks = (h264e_hip_task_t *)calloc((size_t)c->ring, sizeof(*c->tasks));
    c->used = (int32_t (*)[2])calloc((size_t)c->ring, sizeof(int32_t[2]));
    if (!c->traj || !c->tasks || !c->used) { free(c->traj); free(c->tasks); free(c->used); free(c); return -1; }
    /* the ring is sized for speed, not for need: when the device (or the pinned host memory) cannot spare that much, halve it down
     * to the pipeline depth before giving up */
    while (h264e_hip_pool_create(&c->pool, par->device, par->width, par->height, c->ring, c->resident, 1))
    {
        if (c->ring <= c->launch_base + 1 || par->max_chains > 0)
        {
            free(c->traj); free(c->tasks); free(c->used); free(c);
            return -1;
        }
        c->ring = imax(c->launch_base + 1, c->ring/2);
    }
    if (par->keep_records)
    {
        c->rec_store = (h264e_hip_mbrec_t **)calloc((size_t)nframes, sizeof(*c->rec_store));
        c->used_store = (int32_t (*)[2])calloc((size_t)nframes, sizeof(int32_t[2]));
        c->permb_store = (int32_t **)calloc((size_t)nframes, sizeof(*c->permb_store));
        if (!c->rec_store || !c->used_store || !c->permb_store) { H264E_clip_close(c); return -1; }
    }
    H264E_clip_rewind(c);
    *out = c;
    return 0;
}

void H264E_clip_close(H264E_clip_t *c)
{
    if (!c) return;
    h264e_hip_pool_destroy(c->pool);
    if (c->rec_store) { int f; for (f = 0; f < c->nframes; f++) { free(c->rec_store[f]); if (c->permb_store) free(c->permb_store[f]); } }
    free(c->rec_store); free(c->used_store); free(c->permb_store);
    free(c->first_arr); free(c->traj); free(c->tasks); free(c->used); free(c->big);
    free(c);
}

/* frames [first, first + n) of the stream into their ring slots; a slot may only be overwritten once its old frame is encoded */
static int clip_put(H264E_clip_t *c, int first, int n, const uint8_t *i420, int async)
{
    const size_t fsz = (size_t)c->seq.width*c->seq.height*3/2;
    int done = 0;
    if (!c || first < 0 || n < 0 || first + n > c->nframes) { snprintf(g_host_err, sizeof(g_host_err), "upload: bad frame range"); return -1; }
    if (first + n - c->resident > c->next) { snprintf(g_host_err, sizeof(g_host_err), "upload: input ring full (frames %d.. are not encoded yet)", c->next); return -1; }
    while (done < n)
    {
        const int slot = (first + done) % c->resident, run = imin(n - done, c->resident - slot);
        if (async ? h264e_hip_upload_i420_async(c->pool, slot, run, i420 + fsz*(size_t)done) : h264e_hip_upload_i420(c->pool, slot, run, i420 + fsz*(size_t)done))
        {
            g_host_err[0] = 0;          /* H264E_last_error() -> the device layer's text */
            return -1;
        }
        done += run;
    }
    if (async) { if (first + n > c->pending_avail) c->pending_avail = first + n; }
    else if (first + n > c->avail) c->avail = first + n;
    return 0;
}

int H264E_clip_upload(H264E_clip_t *c, int first, int nframes, const uint8_t *i420)
{
    if (clip_put(c, first, nframes, i420, 0)) return -1;
    return h264e_hip_sync(c->pool);
}

/* the same from pinned host memory (H264E_clip_host_alloc) on the pool's copy stream: returns at once, the frames count as
 * uploaded after H264E_clip_upload_wait() */
int H264E_clip_upload_async(H264E_clip_t *c, int first, int nframes, const uint8_t *pinned_i420) { return clip_put(c, first, nframes, pinned_i420, 1); }
int H264E_clip_upload_wait(H264E_clip_t *c)
{
    if (!c || h264e_hip_upload_wait(c->pool)) return -1;
    if (c->pending_avail > c->avail) c->avail = c->pending_avail;
    return 0;
}
/* 1 = every asynchronous upload has landed (the frames now count as uploaded), 0 = still copying */
int H264E_clip_upload_poll(H264E_clip_t *c)
{
    int busy;
    if (!c) return -1;
    busy = h264e_hip_upload_busy(c->pool);
    if (busy < 0) { g_host_err[0] = 0; return -1; }     /* the copy stream failed: H264E_last_error() carries the HIP text */
    if (busy) return 0;
    if (c->pending_avail > c->avail) c->avail = c->pending_avail;
    return 1;
}
void H264E_clip_position(const H264E_clip_t *c, int *next_frame, int *uploaded_frames)
{
    if (next_frame) *next_frame = c ? c->next : 0;
    if (uploaded_frames) *uploaded_frames = c ? c->avail : 0;
}
void H264E_clip_set_idle_hook(H264E_clip_t *c, void (*hook)(void *token), void *token) { if (c) { c->idle_hook = hook; c->idle_token = token; } }
void *H264E_clip_host_alloc(size_t bytes) { return h264e_hip_host_alloc(bytes); }
void H264E_clip_host_free(void *p) { h264e_hip_host_free(p); }

int H264E_clip_generate_synth(H264E_clip_t *c, int first, int nframes, int t0, uint32_t seed)
{
    int done = 0;
    if (!c || first < 0 || nframes < 0 || first + nframes > c->nframes || first + nframes - c->resident > c->next) return -1;
    while (done < nframes)
    {
        const int slot = (first + done) % c->resident, run = imin(nframes - done, c->resident - slot);
        if (h264e_hip_generate_synth(c->pool, slot, run, t0 + done, seed)) return -1;
        done += run;
    }
    if (first + nframes > c->avail) c->avail = first + nframes;
    return h264e_hip_sync(c->pool);
}

/* reconstruction of an already encoded frame (coded size, packed I420), while its picture slot has not been reused: the last
 * ring - 1 frames */
/* the resident input frames back (measurement helper for bench.py's PCIe-inclusive pass) */
int H264E_clip_download(H264E_clip_t *c, int first, int nframes, uint8_t *i420)
{
    if (!c || c->resident < c->nframes) return -1;
    return h264e_hip_download_i420(c->pool, first, nframes, i420);
}

int H264E_clip_read_recon(H264E_clip_t *c, int frame, uint8_t *dst)
{
    if (!c || !dst || frame < 0 || frame >= c->next || frame < c->next - (c->ring - 1)) return -1;
    /* rate control: the launches' hedge leaves (alternative QPs of frames in flight) are encoded in spare slots of the ring, i.e. over
     * the pictures of older frames -- only the frames accepted since the last launch's first frame are guaranteed to be intact */
    if (c->par.kbps > 0 && frame < c->recon_floor) return -1;
    return h264e_hip_read_recon_slot(c->pool, frame % c->ring, dst);
}

/* [3] sums of squared differences (Y, U, V; input vs reconstruction) per frame encoded by the following H264E_clip_encode calls,
 * computed on the device (encode_app --psnr); NULL switches it off */
void H264E_clip_set_ssd_output(H264E_clip_t *c, uint64_t *ssd) { if (c) c->ssd_out = ssd; }

/*
 * GOP sharding of ONE stream: a shard that started from a SPECULATED mv_clusters state is checked once the
 * exact state in front of its first frame is known (the end state of the shard before it).  Walks every accepted frame from
 * exact_in with the records kept by keep_records and compares the rounded candidates each macroblock consumed with the exact
 * ones.  All equal: *restart_frame = -1, the shard's bytes stand, end_state = the exact state behind its last
 * frame (which becomes the encoder's own state).  Otherwise *restart_frame = first frame of the GOP that holds the first
 * divergent macroblock and restart_state = the exact state in front of it: everything from there on has to be encoded again
 * (H264E_clip_restart, then H264E_clip_encode); the frames before it stand.
 */
int H264E_clip_revalidate(H264E_clip_t *c, const int32_t exact_in[2], int *restart_frame, int32_t restart_state[2], int32_t end_state[2])
{
    const int nslices = c && c->par.slices > 1 ? imin(imin(c->par.slices, H264E_HIP_MAX_SLICES), c->seq.nmby) : 1;
    int32_t s[2], gop_state[2];
    int f;
    if (!c || !c->rec_store || !exact_in || !restart_frame) return -1;
    s[0] = gop_state[0] = exact_in[0]; s[1] = gop_state[1] = exact_in[1];
    *restart_frame = -1;
    for (f = 0; f < c->next; f++)
    {
        if (f % c->gop_len == 0) { gop_state[0] = s[0]; gop_state[1] = s[1]; }
        if (!c->rec_store[f]) return -1;
        if (clusters_walk(s, c->rec_store[f], c->seq.nmbx, c->seq.nmby, nslices, c->permb_store[f] ? c->permb_store[f] : c->used_store[f], c->permb_store[f] != NULL, NULL) >= 0)
        {
            *restart_frame = f - f % c->gop_len;
            if (restart_state) { restart_state[0] = gop_state[0]; restart_state[1] = gop_state[1]; }
            return 0;
        }
    }
    if (end_state) { end_state[0] = s[0]; end_state[1] = s[1]; }
    c->state[0] = s[0]; c->state[1] = s[1];
    return 0;
}

/* the kept macroblock records of an accepted frame (keep_records): {mv[0] packed (y << 16) | (x & 0xffff), type -1 skip / 0..3 inter
 * partitioning / 5 I4x4 / 6 I16x16, consumed-the-cluster-candidates flag} per macroblock -- what a per-macroblock comparison with another
 * implementation's trace needs (tests) */
int H264E_clip_read_records(H264E_clip_t *c, int frame, void *dst /* nmb x 8 bytes */)
{
    if (!c || !dst || !c->rec_store || frame < 0 || frame >= c->next || !c->rec_store[frame]) return -1;
    memcpy(dst, c->rec_store[frame], sizeof(h264e_hip_mbrec_t)*(size_t)c->seq.nmb);
    return 0;
}

/* continue (again) at `frame`, which must start a GOP, with `state` in front of it; the frames behind it are encoded again by the
 * next H264E_clip_encode calls (their inputs must be resident / uploaded again) */
int H264E_clip_restart(H264E_clip_t *c, int frame, const int32_t state[2])
{
    if (!c || !state || frame < 0 || frame > c->avail || frame % c->gop_len) return -1;
    c->next = frame;
    if (c->resident < c->nframes) c->avail = c->pending_avail = frame;      /* a ring: the inputs from here on have to be uploaded again */
    c->state[0] = state[0]; c->state[1] = state[1];
    free(c->first_arr); c->first_arr = NULL; c->first_dev = 0;
    c->first_row = 0; c->have_after = 0;
    c->rc_frame = -1;
    return 0;
}

/* diagnostic (stamps build): per-phase cycle sums since the last call */
int H264E_clip_stamps(H264E_clip_t *c, unsigned long long *dst) { return c ? h264e_hip_stamps_read(c->pool, dst, 1) : -1; }

int H264E_clip_encode(H264E_clip_t *c, uint8_t *out, size_t cap, size_t *out_bytes, int *frame_bytes, int profile, H264E_clip_stats_t *st)
{
    const int nmb = c->seq.nmb, G = c->gop_len, K = c->ring, no_deblock = (c->par.speed == 8 || c->par.speed == 10);
    const int nslices = c->par.slices > 1 ? imin(imin(c->par.slices, H264E_HIP_MAX_SLICES), c->seq.nmby) : 1;
    /* frame-level rate control (encode_app --kbps, minih264e_test.c:596-600: desired_frame_bytes = kbps*1000/8/30, QP 10..50):
     * a frame's QP is a function of the byte count of the frame before it (h264-lab.h:5924-6141) and moves almost every
     * frame: the frames behind the first one of a launch run on a speculated QP (see the launch loop), the controller on the host */
    const int rc_on = c->par.kbps > 0, desired_frame_bytes = c->par.kbps*1000/8/30, qp_min = rc_on ? 10 : c->par.qp, qp_max = rc_on ? 50 : c->par.qp;
    const int pic_init_qp = imax(imin(30, qp_max), qp_min);     /* h264-lab.h:6768-6770 */
    const int idr_state = c->par.first_idr_pic_id_state & 1;
    const int first = c->next;
    h264e_hip_task_t *tasks = c->tasks;
    int32_t (*used)[2] = c->used;
    long long far_reads = 0;
    uint16_t qdat_i[2][42], qdat_p[2][42];
    size_t pos = 0;
    int rc = -1, i, full = 0, qp = c->rc_qp, spin_retries = 0;
    H264E_clip_stats_t stats;
    double t0;
    memset(&stats, 0, sizeof(stats));
    g_host_err[0] = 0;
    build_qdat(qdat_i, qp, 0);
    build_qdat(qdat_p, qp, 1);
    h264e_hip_profile(c->pool, profile);
    stats.chains = K - 1;
    stats.first_frame = first;

    while (c->next < c->avail && !full)
    {
        const int n = c->next, limit = c->avail;       /* frames uploaded from the idle hook during this launch join the next one */
        /* with --psnr style statistics every frame's picture must still be in its slot when the launch has drained */
        /* rate control: frame n+1's QP is a function of frame n's size (h264-lab.h:5924-6141), so the frames behind the first one of
         * a launch run on a SPECULATED QP: the controller is run ahead on predicted sizes (the last frame of the same kind); when a
         * frame's real size is in, the exact QP of the next one is computed and compared with what it was given -- a mismatch stops
         * the launch there (measured hit rates of the one-ahead guess: 15-66 %) */
        const int rc_depth = imax(1, imin(8, getenv("H264E_RC_DEPTH") ? atoi(getenv("H264E_RC_DEPTH")) : (nmb >= 60000 ? 3 : 6)));     /* measured optima: 8K 3, below 6 */
        const int F = rc_on ? imax(1, imin(imin(K - 1, rc_depth), limit - n)) : imin(imin(K - 1, c->launch_frames), limit - n);
        int qp_task[8];
        rc_t rc_ahead;
        int nvalid = 0;
        t0 = now_ms();
        memset(tasks, 0, sizeof(*tasks)*(size_t)K);
        if (rc_on) qp = c->rc_qp;               /* (the task loop of the previous launch left a speculated value here) */
        if (rc_on && c->rc_frame != n)
        {
            const int key = (n % G) == 0;
            qp = c->rc_qp = rc_frame_start(&c->rcs, c->par.gop, nmb, c->par.vbv_size_bytes, desired_frame_bytes, qp_min, qp_max, key);
            build_qdat(key ? qdat_i : qdat_p, qp, !key);
            c->rc_frame = n;
        } else if (rc_on)
            build_qdat((n % G) == 0 ? qdat_i : qdat_p, qp, (n % G) != 0);
        rc_ahead = c->rcs;
        for (i = 0; i < F; i++)
        {
            h264e_hip_task_t *t = tasks + i;
            const int f = n + i, key = (f % G) == 0;
            if (rc_on && i > 0)
            {
                /* the controller, run ahead: the frame in front is predicted to weigh what the last frame of its kind did */
                const int pkey = ((f - 1) % G) == 0, pred = c->rc_last_bytes[pkey] > 0 ? c->rc_last_bytes[pkey] : desired_frame_bytes;
                rc_frame_end(&rc_ahead, nmb, c->par.vbv_size_bytes, desired_frame_bytes, pred, pkey, 0);
                qp = rc_frame_start(&rc_ahead, c->par.gop, nmb, c->par.vbv_size_bytes, desired_frame_bytes, qp_min, qp_max, key);
                build_qdat(key ? qdat_i : qdat_p, qp, !key);
            }
            qp_task[i & 7] = qp;
            t->active = 1; t->frame_index = f % c->resident; t->frame_slot = 0;
            t->slice_type = key ? SLICE_I : SLICE_P;
            t->qp = qp; t->speed = c->par.speed;
            /* frame_num restarts at every key frame; idr_pic_id toggles with every key frame (h264-lab.h:6774-6775) */
            t->nslices = nslices;
            slice_header_bits(&c->seq, key, f % G, (idr_state ^ ((f/G + 1) & 1)), qp, pic_init_qp, no_deblock, nslices, &t->hdr_nal, &t->hdr_bits, &t->hdr_nbits);
            memcpy(t->qdat, key ? qdat_i : qdat_p, sizeof(t->qdat));
            t->stream_mode = 1; t->slot = f % K;
            t->ref_slot = key ? -1 : (f - 1) % K;
            t->ref_in_flight = !key && i > 0;
            /* frame 0 of the launch gets the exact state; the frames behind it the best prediction of what it leaves */
            t->mv_clusters[0] = used[i][0] = (i && c->have_after) ? c->after[0] : c->state[0];
            t->mv_clusters[1] = used[i][1] = (i && c->have_after) ? c->after[1] : c->state[1];
            /* validation on the device: the frame's finalizer walks its records exactly (task 0 from the exact state, the others from
             * the verdict in front of them) and stops the launch itself when a macroblock consumed the wrong candidates */
            t->walk_on_device = 1;
            t->exact_state[0] = c->state[0]; t->exact_state[1] = c->state[1];
            t->mv_clusters_per_mb = NULL;
            t->traj_from_device = (i == 0) && c->first_dev;
            t->first_row = (i == 0 && c->first_dev) ? c->first_row : 0;
            t->narrow_window = c->narrow;
        }
        /* Hedges (rate control): the speculated QP of a frame is usually off by one or two when it is off, so every frame behind the
         * first one is ALSO encoded with the neighbouring QPs, as leaves in spare slots (same reference, same mv_clusters
         * speculation; the chip is nearly empty in this mode).  When the chain breaks at a frame, the leaf with the exact QP -- if
         * there is one -- is the frame; its picture then moves to the slot the frame number owns and the launch ends there. */
        int nh = 0, hedge_level[32], hedge_qp[32];
        if (rc_on && F > 1)
        {
            static const int dq[4] = { -1, 1, -2, 2 };
            const int rc_hedge = imax(0, imin(4, getenv("H264E_RC_HEDGE") ? atoi(getenv("H264E_RC_HEDGE")) : (nmb >= 60000 ? 0 : nmb >= 20000 ? 2 : 4)));      /* measured: 8K is bound by the resident workgroups, leaves only cost there */
            int k, a;
            for (k = 1; k < F; k++)
                for (a = 0; a < rc_hedge; a++)
                {
                    const int q = qp_task[k & 7] + dq[a], f = n + k, key = (f % G) == 0, j = F + nh;
                    h264e_hip_task_t *t = tasks + j;
                    uint16_t qd[2][42];
                    if (q < qp_min || q > qp_max || nh >= 32 || j + 2 >= K) continue;
                    *t = tasks[k];
                    build_qdat(qd, q, !key);
                    t->qp = q;
                    memcpy(t->qdat, qd, sizeof(t->qdat));
                    slice_header_bits(&c->seq, key, f % G, (idr_state ^ ((f/G + 1) & 1)), q, pic_init_qp, no_deblock, nslices, &t->hdr_nal, &t->hdr_bits, &t->hdr_nbits);
                    t->slot = (n + j) % K;              /* a slot no frame of this launch owns */
                    t->walk_parent = k;                 /* = index of the chain's frame in front of it, + 1 */
                    t->walk_quiet = 1;                  /* its own validation failing stops nobody else */
                    used[j][0] = used[k][0]; used[j][1] = used[k][1];
                    hedge_level[nh] = k; hedge_qp[nh] = q; nh++;
                }
        }
        if (nh) c->recon_floor = n;             /* the leaves' slots held the pictures of frames n - K + F .. : gone now */
        stats.rounds++;
        const int after_stop = c->stopped_before;   /* this launch follows one that was stopped (mis-speculation, rate-control miss): its first frame pays the refill */
        c->have_after = 0;
        const double t_submit = now_ms();
        double t_first = 0, t_last = 0;
        int rc_miss = 0, take = -1, moved_from = -1, moved_to = -1, relaunch = 0;
        const char *why = "all frames delivered";       /* what ended the launch (H264E_DEBUG) */
        if (h264e_hip_submit(c->pool, tasks)) goto done;

        /* consume the frames in stream order while the launch is still running */
        for (i = 0; i < F || take >= 0; i++)
        {
            const int is_hedge = take >= 0, ti = is_hedge ? take : i;
            const int f = is_hedge ? n + hedge_level[take - F] : n + i, key = (f % G) == 0, slot = tasks[ti].slot, per_mb = (ti == 0 && c->first_dev);
            h264e_hip_result_t r1;
            int dn, idle = 0;
            int32_t cc[2];
            memset(&r1, 0, sizeof(r1));
            while ((dn = h264e_hip_stream_done(c->pool, slot, &r1)) == 0)
            {
                if (!h264e_hip_busy(c->pool) && ++idle > 2) break;      /* the launch ended without finishing this job */
                if (c->idle_hook) c->idle_hook(c->idle_token);
                sched_yield();
            }
            take = -1;
            if (dn == 0) dn = h264e_hip_stream_done(c->pool, slot, &r1);
            if (is_hedge && dn != 1)
            {
                /* the leaf did not make it (its own mv_clusters validation failed, or the launch was stopped): the frame is simply
                 * encoded in the next launch */
                if (dn < 0) goto done;
                if (h264e_hip_stream_abort(c->pool)) goto done;
                why = "the hedge leaf with the exact QP did not complete (its own mv_clusters validation, or stopped)";
                break;
            }
            if (dn == 2 && r1.walk_status == 2)
            {
                /* the device's exact walk found a macroblock of this frame that consumed other rounded candidates than the exact
                 * ones and stopped the launch.  Every macroblock before first_bad consumed the right ones: its row and
                 * the rows above are bit-identical in the next encode and are kept; the frame goes again from that row with the
                 * walked per-macroblock trajectory (it stays on the device), the frames behind it with the walk's end state */
                c->first_row = r1.first_bad/c->seq.nmbx;
                c->first_dev = 1;
                c->after[0] = r1.state_out[0]; c->after[1] = r1.state_out[1]; c->have_after = 1;
                stats.reencoded_gops++;             /* counts relaunches */
                why = "mv_clusters mis-speculation";
                break;
            }
            if (dn != 1)
            {
                if (dn < 0) goto done;
                /* The frame did not complete and nobody asked it to stop.  When the kernel reports that a bounded wait expired -- workgroups
                 * starved of wave slots, e.g. by another process' launch on the same device -- nothing wrong was returned: every frame
                 * accepted so far stands, and the unfinished ones are simply launched again (a fresh launch in this process).  Anything
                 * else, and a wait that keeps expiring without a single frame getting through, is an error. */
                if (h264e_hip_sync(c->pool) && strstr(h264e_hip_last_error(), "bounded spin expired") && spin_retries < 3)
                {
                    spin_retries++;
                    stats.spin_relaunches++;
                    relaunch = 1;
                    break;
                }
                if (!g_host_err[0] && !h264e_hip_last_error()[0]) snprintf(g_host_err, sizeof(g_host_err), "frame %d did not complete", f);
                goto done;
            }
            t_last = now_ms();
            if (i == 0) { t_first = t_last; far_reads = 0; }
            far_reads += r1.far_reads;
            if (r1.overflow) { snprintf(g_host_err, sizeof(g_host_err), "bit buffer overflow (frame %d)", f); (void)h264e_hip_stream_abort(c->pool); (void)h264e_hip_sync(c->pool); goto done; }
            cc[0] = r1.state_out[0]; cc[1] = r1.state_out[1];      /* exact state behind this frame (device walk), committed once the frame is accepted */
            t0 = now_ms();
            {
                /* the frame as the kernel exported it: complete NALs (start codes and emulation prevention done on the device) */
                const uint8_t *nals = h264e_hip_stream_rbsp(c->pool, slot);
                size_t start = pos, w = 0, need = (key ? 64 : 0) + r1.nbytes;
                if (r1.in_device)
                {
                    /* larger than the host-mapped mirror (sized for ordinary frames): copy it from the slot's device NAL arena */
                    if (c->big_cap < r1.nbytes) { free(c->big); c->big_cap = (size_t)r1.nbytes*2; c->big = (uint8_t *)malloc(c->big_cap); }
                    if (!c->big || h264e_hip_stream_fetch_nals(c->pool, slot, c->big, r1.nbytes)) { c->big_cap = 0; (void)h264e_hip_stream_abort(c->pool); (void)h264e_hip_sync(c->pool); goto done; }
                    nals = c->big;
                }
                if (pos + need > cap)
                {
                    /* the caller's buffer is full: stop here, the stream continues with this frame in the next call */
                    if (h264e_hip_stream_abort(c->pool)) goto done;
                    full = 1;
                    break;
                }
                if (key) pos += write_sps_pps(&c->seq, pic_init_qp, out + pos, NULL, NULL);
                w = emit_slices(out + pos, cap - pos, nals, &r1, NULL, NULL);
                if (!w) { snprintf(g_host_err, sizeof(g_host_err), "malformed frame export"); (void)h264e_hip_stream_abort(c->pool); (void)h264e_hip_sync(c->pool); goto done; }
                pos += w;
                if (frame_bytes) frame_bytes[f - first] = (int)(pos - start);
                if (rc_on)
                {
                    rc_frame_end(&c->rcs, nmb, c->par.vbv_size_bytes, desired_frame_bytes, (int)(pos - start), key, r1.all_skipped);
                    c->rc_last_bytes[key] = (int)(pos - start);
                    if (f + 1 < c->nframes)
                    {
                        /* the exact QP of the next frame; a frame of this launch that was given another one is stopped */
                        const int nkey = ((f + 1) % G) == 0;
                        qp = c->rc_qp = rc_frame_start(&c->rcs, c->par.gop, nmb, c->par.vbv_size_bytes, desired_frame_bytes, qp_min, qp_max, nkey);
                        c->rc_frame = f + 1;
                        if (is_hedge) rc_miss = 1;                  /* a leaf has nothing behind it */
                        else if (i + 1 < F && qp_task[(i + 1) & 7] != qp)
                        {
                            int hh;
                            rc_miss = 1;
                            for (hh = 0; hh < nh; hh++) if (hedge_level[hh] == i + 1 && hedge_qp[hh] == qp) { take = F + hh; rc_miss = 0; }
                        }
                    } else if (is_hedge) rc_miss = 1;
                }
            }
            if (c->rec_store)
            {
                /* what this accepted frame consumed: records + the candidates it was given (H264E_clip_revalidate) */
                const size_t rb = sizeof(h264e_hip_mbrec_t)*(size_t)nmb;
                if (!c->rec_store[f]) c->rec_store[f] = (h264e_hip_mbrec_t *)malloc(rb);
                if (!c->rec_store[f]) goto done;
                memcpy(c->rec_store[f], h264e_hip_stream_mbrec(c->pool, slot), rb);
                c->used_store[f][0] = used[ti][0]; c->used_store[f][1] = used[ti][1];
                free(c->permb_store[f]); c->permb_store[f] = NULL;
                if (per_mb)
                {
                    c->permb_store[f] = (int32_t *)malloc(sizeof(int32_t)*2*(size_t)nmb);
                    if (!c->permb_store[f] || h264e_hip_stream_fetch_traj(c->pool, slot, 1, c->permb_store[f])) goto done;
                }
            }
            c->state[0] = cc[0]; c->state[1] = cc[1];
            if (ti == 0) c->first_dev = 0;
            stats.assemble_ms += now_ms() - t0;
            nvalid++;
            if (is_hedge) { moved_from = slot; moved_to = f % K; }
            if (rc_miss)
            {
                if (h264e_hip_stream_abort(c->pool)) goto done;
                stats.reencoded_gops++;
                why = is_hedge ? "QP miss covered by a hedge leaf (nothing behind a leaf)" : "QP miss, no hedge leaf with the exact QP";
                break;
            }
            if (take >= 0) i = F - 1;           /* the chain ends here: one more round, for the leaf */
        }
        if (relaunch) { (void)h264e_hip_stream_abort(c->pool); (void)h264e_hip_sync(c->pool); }       /* (the failure was reported above; the launch is over) */
        else if (h264e_hip_sync(c->pool)) goto done;    /* the launch has drained (immediately after an abort) */
        if (nvalid) spin_retries = 0;
        if (moved_from >= 0 && h264e_hip_stream_copy_picture(c->pool, moved_from, moved_to)) goto done;
        stats.encode_ms += now_ms() - t_submit;
        if (c->ssd_out && nvalid)
        {
            /* device-side sums of squared differences of the frames just validated: their pictures are still in their slots */
            if (h264e_hip_ssd_frames(c->pool, nvalid, n % c->resident, c->resident, n % K, K, c->ssd_out + 3*(size_t)(n - first))) goto done;
        }
        c->next = n + nvalid;
        if (after_stop && nvalid) { stats.relaunches_timed++; stats.first_frame_ms_after_relaunch += t_first - t_submit; }
        c->stopped_before = nvalid < F && !full;
        /* frames per launch: back to the pipeline depth after a mis-speculation, twice as many after a clean launch */
        /* (after an event: twice the frames the stopped launch got through, an estimate of the spacing of the events) */
        if (nvalid < F && !full) c->launch_frames = imin(imax(c->launch_base, 2*nvalid), K - 1);
        else if (nvalid == F) c->launch_frames = imin(2*c->launch_frames, K - 1);
        /* Window geometry for the next launch.  More than one macroblock in eight leaving the narrow window over at least 8 frames:
         * the wide one pays -- for a while: such motion is often a transient (a scene cut, an object wrapping around), so the narrow
         * geometry is tried again after wide_hold frames, a spell that doubles every time it fails again. */
        if (c->narrow)
        {
            c->far_acc += far_reads; c->far_frames += nvalid;
            if (c->far_frames >= 8)
            {
                if (c->far_acc > (long long)c->far_frames*nmb/8)
                {
                    c->narrow = 0;
                    c->wide_until = c->next + c->wide_hold;
                    c->wide_hold = imin(c->wide_hold*2, 960);
                } else if (c->far_frames >= 64) c->wide_hold = 30;
                c->far_acc = 0; c->far_frames = 0;
            }
        } else if (c->narrow_ok && c->next >= c->wide_until) { c->narrow = 1; c->far_acc = 0; c->far_frames = 0; }
        if (getenv("H264E_DEBUG"))
        {
            const double t_end = now_ms();
            if (!nvalid) t_first = t_last = t_end;              /* no frame of it was delivered: "first frame after" = when its verdict was in */
            fprintf(stderr, "clip launch %d (first row %d, %s window, %lld far reads): %d frames in flight, %d valid, next %d; first frame after %.2f ms, then %.3f ms/frame, drained %.2f ms after the last; ended by: %s%s\n",
                    stats.rounds, tasks[0].first_row, tasks[0].narrow_window ? "narrow" : "wide", far_reads, F, nvalid, c->next, t_first - t_submit, nvalid > 1 ? (t_last - t_first)/(nvalid - 1 + (nvalid < F)) : 0.0, t_end - t_last,
                    full ? "output buffer full" : why, nh ? " (+ hedge leaves)" : "");
        }
        if (full && c->next == first)
        {
            snprintf(g_host_err, sizeof(g_host_err), "output buffer too small for one frame");
            goto done;
        }
    }
    stats.frames = c->next - first;
    stats.delivered_mbs = (long long)stats.frames*nmb;
    {
        unsigned long long pm = 0;
        if (!h264e_hip_mb_counter(c->pool, &pm, 1)) stats.processed_mbs = (long long)pm;
    }
    stats.mv_clusters_out[0] = c->state[0]; stats.mv_clusters_out[1] = c->state[1];
    stats.next_idr_pic_id_state = idr_state ^ (((c->next + G - 1)/G) & 1);
    h264e_hip_profile_read(c->pool, &stats.mb_kernel_ms, &stats.splice_kernel_ms, &stats.kernel_launches);
    if (out_bytes) *out_bytes = pos;
    rc = 0;
done:
    if (rc) h264e_hip_release(c->pool);         /* a failure between submit and sync: give the device's launch lock back */
    if (st) *st = stats;
    return rc;
}


/* ------------------------------------------------------------------ several clips of one picture size on ONE device at once */

typedef struct
{
    H264E_clip_t *clip; h264e_hip_group_t *group;
    uint8_t *out; size_t cap; size_t *out_bytes; int *frame_bytes; H264E_clip_stats_t *st;
    int rc; char err[256];
} multi_arg_t;

static void *multi_thread(void *p)
{
    multi_arg_t *a = (multi_arg_t *)p;
    a->rc = H264E_clip_encode(a->clip, a->out, a->cap, a->out_bytes, a->frame_bytes, 0, a->st);
    if (a->rc) snprintf(a->err, sizeof(a->err), "%s", H264E_last_error());
    h264e_hip_group_leave(a->group, a->clip->pool);     /* the others no longer wait for this one's launches */
    return NULL;
}

/*
 * A single-slice stream spends most of a pass in pipeline drains and refills (one per mis-speculated mv_clusters state);
 * independent streams fill each other's gaps when their frames share ONE launch (include/h264e_hip.h launch groups): every clip is
 * encoded by its own host thread exactly like H264E_clip_encode does, the device layer merges the threads' launches round by round.
 * Same bytes per clip as H264E_clip_encode.
 */
int H264E_clip_encode_multi(H264E_clip_t **clips, int nclips, uint8_t **out, const size_t *cap, size_t *out_bytes, int **frame_bytes, H264E_clip_stats_t *stats)
{
    multi_arg_t *a;
    pthread_t *th;
    int i, rc = 0, done = 0, saved_base[8] = { 0 };
    g_host_err[0] = 0;
    if (!clips || nclips <= 0 || nclips > 8 || !out || !cap || !out_bytes) { snprintf(g_host_err, sizeof(g_host_err), "clip_encode_multi: bad argument"); return -1; }
    for (i = 0; i < nclips; i++) if (!clips[i]) { snprintf(g_host_err, sizeof(g_host_err), "clip_encode_multi: clip %d is NULL", i); return -1; }
    if (nclips == 1) return H264E_clip_encode(clips[0], out[0], cap[0], &out_bytes[0], frame_bytes ? frame_bytes[0] : NULL, 0, stats);
    a = (multi_arg_t *)calloc((size_t)nclips, sizeof(*a));
    th = (pthread_t *)calloc((size_t)nclips, sizeof(*th));
    if (!a || !th) { free(a); free(th); snprintf(g_host_err, sizeof(g_host_err), "out of host memory"); return -1; }
    /* one launch group per batch: a group holds as many streams of this picture size as one grid can carry safely (h264e_pool.h
     * h264e_hip_group_join: 5 at 1080p, 2 at 4K, 1 at 8K); the clips that do not fit go in the next batch */
    while (done < nclips && !rc)
    {
        h264e_hip_group_t *g = NULL;
        int first = done, n = 0, started = 0;
        if (h264e_hip_group_create(&g, clips[first]->par.device)) { snprintf(g_host_err, sizeof(g_host_err), "clip_encode_multi: %s", h264e_hip_last_error()); rc = -1; break; }
        while (first + n < nclips && !h264e_hip_group_join(g, clips[first + n]->pool)) n++;
        if (!n) { snprintf(g_host_err, sizeof(g_host_err), "clip_encode_multi: clip %d cannot join a launch group (%s)", first, h264e_hip_last_error()); h264e_hip_group_destroy(g); rc = -1; break; }
        for (i = first; i < first + n; i++)
        {
            /* The members' launches run in lock step, so a member that was stopped by a mis-speculation idles until the round's longest
             * launch is over: streams with frequent events (single slice, constant QP) get shorter launches inside a group -- a third of
             * the pipeline depth -- which costs streams WITHOUT staggered events a few percent and gives staggered ones 13 %
             * (measured, 4 different 1080p clips: 12.8 -> 14.6 M MB/s aggregate; 4 identical ones 21.4 -> 20.7 M) */
            saved_base[i] = clips[i]->launch_base;
            if (n > 1 && clips[i]->par.slices <= 1 && clips[i]->gop_len > 1 && clips[i]->par.kbps == 0)
            {
                clips[i]->launch_base = imax(8, clips[i]->launch_base/3);
                clips[i]->launch_frames = imin(clips[i]->launch_frames, clips[i]->launch_base);
            }
            a[i].clip = clips[i]; a[i].group = g; a[i].out = out[i]; a[i].cap = cap[i]; a[i].out_bytes = &out_bytes[i];
            a[i].frame_bytes = frame_bytes ? frame_bytes[i] : NULL; a[i].st = stats ? &stats[i] : NULL;
            if (pthread_create(&th[i], NULL, multi_thread, &a[i])) { snprintf(g_host_err, sizeof(g_host_err), "clip_encode_multi: cannot start a thread"); rc = -1; break; }
            started++;
        }
        /* a clip whose thread never started must not be waited for by the others */
        for (i = first + started; i < first + n; i++) h264e_hip_group_leave(g, clips[i]->pool);
        for (i = first; i < first + started; i++)
        {
            pthread_join(th[i], NULL);
            if (a[i].rc && !rc) { rc = a[i].rc; snprintf(g_host_err, sizeof(g_host_err), "clip %d: %s", i, a[i].err); }
        }
        for (i = first; i < first + n; i++) if (saved_base[i] > 0) clips[i]->launch_base = saved_base[i];
        h264e_hip_group_destroy(g);
        done = first + n;
    }
    free(a); free(th);
    return rc;
}
