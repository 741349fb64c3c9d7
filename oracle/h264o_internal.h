/*
 * h264o_internal.h -- CPU restatement of the h264-lab per-frame encode path.
 *
 * TEST INFRASTRUCTURE: this is the oracle the HIP path is checked against.  It
 * is a from-scratch scalar C restatement of the algorithm in
 * /root/reference/src/h264-lab.h ("H:" below), pinned against the compiled
 * reference (oracle/_ref) by tests/test_oracle_*.py.  Nothing under
 * h264-lab_amd/ may include, link or call it.
 *
 * Coefficient blocks are stored the way the reference stores them
 * (H:2385-2409): index = 4*k_h + k_v ("transposed"), and CAVLC scans that
 * index linearly 15..0 (H:2786-2798) -- there is no zig-zag.
 */
#ifndef H264O_INTERNAL_H
#define H264O_INTERNAL_H

#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include "h264o.h"

#define MV_NA       0x8000          /* H:3200 */
#define NNZ_NA      64              /* H:3206 */
#define AV_T 1                      /* H:511-514 */
#define AV_L 2
#define AV_TL 4
#define AV_TR 8

#define QMODE_I4    2               /* H:505-508 */
#define QMODE_INTER 8
#define QMODE_I16   9
#define QMODE_CHROMA 5

/* qdat layout, H:701-728 */
#define QD_RND 6
#define QD_THR1 10
#define QD_THR2 18

typedef int32_t mv32;               /* packed (y << 16) | (x & 0xffff), H:3446 */
static inline int mvx(mv32 v) { return (int16_t)(v & 0xffff); }
static inline int mvy(mv32 v) { return (int16_t)((uint32_t)v >> 16); }
static inline mv32 mvmk(int x, int y) { return (mv32)(((uint32_t)y << 16) | ((uint32_t)x & 0xffff)); }
static inline mv32 mvadd(mv32 a, mv32 b) { return mvmk(mvx(a) + mvx(b), mvy(a) + mvy(b)); }
static inline mv32 mvsub(mv32 a, mv32 b) { return mvmk(mvx(a) - mvx(b), mvy(a) - mvy(b)); }
static inline mv32 mvround(mv32 a) { return mvmk((mvx(a) + 1) & ~3, (mvy(a) + 1) & ~3); } /* H:3498 */

typedef struct { int16_t qv[16]; int16_t dq[16]; } qblk_t;

/* bit writer: MSB first into a byte buffer */
typedef struct { uint8_t *buf; size_t cap; uint64_t acc; int nacc; size_t pos; } bitw_t;
void bw_init(bitw_t *b, uint8_t *buf, size_t cap);
void bw_put(bitw_t *b, int n, uint32_t v);
void bw_ue(bitw_t *b, uint32_t v);
void bw_se(bitw_t *b, int v);
size_t bw_bits(const bitw_t *b);
void bw_flush(bitw_t *b);           /* pads the last byte with zero bits */
int  bits_ue(int v);
int  bits_se(int v);

/* pixel kernels */
int  sad_wh(const uint8_t *a, int as, const uint8_t *b, int bs, int w, int h);
int  sad_16x16_q(const uint8_t *a, int as, const uint8_t *b, int bs, int sad4[4]);
void interp_luma(const uint8_t *ref, int stride, int mvx_abs, int mvy_abs, int w, int h, uint8_t *dst /*stride 16*/);
void interp_chroma(const uint8_t *ref, int stride, int mvx_abs, int mvy_abs, int w, int h, uint8_t *dst /*stride 16*/);
void avg_wh(const uint8_t *a, const uint8_t *b, uint8_t *d, int w, int h);   /* all stride 16 */
void pred16(uint8_t *dst, const uint8_t *left, const uint8_t *top, int mode);
void pred_chroma(uint8_t *dst, const uint8_t *left, const uint8_t *top, int mode);
int  i4_choose(const uint8_t *in, uint8_t *pred, int avail, const uint8_t *top8, const uint8_t *left4, int tl,
               int mpred, int penalty, int *psad);
int  xform_quant(const uint8_t *inp, int is, const uint8_t *pred, int mode, qblk_t *q, int16_t *dc, const uint16_t *qdat);
void quant_luma_dc(qblk_t *q, int16_t *dc, int16_t *lev, const uint16_t *qdat);
int  quant_chroma_dc(qblk_t *q, int16_t *dc, int16_t *lev, const uint16_t *qdat);
void recon_blocks(uint8_t *out, int os, const uint8_t *pred, qblk_t *q, int side, uint32_t mask);
void cavlc_block(bitw_t *b, const int16_t *coef, int first, int maxn, int nctx, uint8_t *nnz_out);
void deblock_mb(uint8_t *y, int ys, uint8_t *u, uint8_t *v, int cs, const uint8_t bs[32],
                int qp, int qp_left, int qp_top);
void build_qdat(uint16_t qdat[2][42], int qp, int p_slice);

#endif
