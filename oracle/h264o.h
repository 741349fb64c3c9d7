/*
 * h264o.h -- public interface of the CPU oracle (TEST INFRASTRUCTURE).
 *
 * A scalar C restatement of the reference encoder's per-frame path
 * (/root/reference/src/h264-lab.h:6654-6861 H264E_encode and everything below
 * it), constant-QP or frame-level rate control, one slice or N row-band slices per frame, one reference frame.  Used only by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the
 * checker for the HIP path; never linked into the product library.
 */
#ifndef H264O_H
#define H264O_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct
{
    int width, height;      /* picture size, even; need not be a multiple of 16 */
    int gop;                /* key-frame period; 0 = only the first frame is a key frame */
    int qp;                 /* constant QP, clamped to [10,51] like the reference (H:6707-6715) */
    int speed;              /* encode_speed: 0 best ... 10 fastest (H:76-78) */
    int vbv_size_bytes;     /* only selects the SPS level here; encode_app passes 12500 (T:524) */
    int kbps;               /* 0 = constant QP; else frame-level rate control like encode_app --kbps (T:596-600) */
    int slices;             /* 0 / 1 = one slice per frame; N > 1 = N row bands, the reference's -DH264E_MAX_THREADS build with
                               --threads N (H:6511-6574): per-band slices with deblock idc 2, mv_clusters restarted per band */
} h264o_param_t;

/* per-macroblock decision record, for debugging a second implementation against the oracle */
typedef struct
{
    int8_t  type;           /* -1 skip, 0..3 inter partitioning, 5 I4x4, 6 I16x16 (H:659) */
    uint8_t cbp;            /* luma | chroma << 4 */
    int16_t mvx, mvy;       /* mv[0], quarter-pel */
    uint32_t bitpos;        /* slice-payload bit position after this macroblock */
} h264o_mbtrace_t;

/* state that crosses a GOP boundary in constant-QP mode */
typedef struct
{
    int32_t mv_clusters[2]; /* packed (y << 16) | (x & 0xffff) */
    int next_idr_pic_id;
} h264o_chain_t;

typedef struct h264o_enc h264o_enc_t;

h264o_enc_t *h264o_open(const h264o_param_t *par);
void h264o_close(h264o_enc_t *e);
/* Encode the next frame.  *out points into encoder-owned memory, valid until the next call. */
int  h264o_encode(h264o_enc_t *e, const uint8_t *const yuv[3], const int stride[3], uint8_t **out, int *out_bytes);
int  h264o_get_qp(const h264o_enc_t *e);        /* QP of the last encoded frame */
/* H:6898-6913 H264E_set_vbv_state (vbv_fullness_bytes < 0: no change) */
void h264o_set_vbv_state(h264o_enc_t *e, int vbv_size_bytes, int vbv_fullness_bytes);
void h264o_get_chain(const h264o_enc_t *e, h264o_chain_t *c);
void h264o_set_chain(h264o_enc_t *e, const h264o_chain_t *c);
/* copy of the last reconstructed (deblocked) frame, I420 w x h of the CODED size; returns coded w/h */
void h264o_get_recon(const h264o_enc_t *e, uint8_t *dst, int *cw, int *ch);
/* per-MB trace of the last frame (nmb entries) */
const h264o_mbtrace_t *h264o_get_trace(const h264o_enc_t *e, int *nmb);
/* Encode nframes of a packed I420 clip; returns bytes written to out (or -1 when cap is too small). */
long h264o_encode_clip(const h264o_param_t *par, const uint8_t *clip, int nframes, uint8_t *out, size_t cap,
                       int *frame_bytes /* optional [nframes] */);

#ifdef __cplusplus
}
#endif
#endif
