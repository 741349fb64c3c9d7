/*
 * h264e_dev.h -- data layout shared by the kernels and their launcher.
 *
 * HBM layout per SLOT ("chain"): plain mode = one sequential stream of frames per slot with two pictures in ping-pong;
 * stream mode = a ring of slots, frame f of the stream in slot f mod K, each referencing the picture of the slot before it:
 *   rec[2][3]   two reconstructed pictures (stream mode uses the first), coded size, no guard band:
 *               out-of-picture reference reads clamp coordinates, which equals the reference's
 *               replicated borders (h264-lab.h:2232-2248, 3580-3596)
 *   bottom      one 64-byte record per macroblock: what the row below needs from it
 *   pend        per macroblock: its bottom lines until the row below has filtered them (h264e_mbpend_t)
 *   progress    one counter per macroblock row: macroblocks finished in that row (wavefront hand-off), followed by one `decided`
 *               counter per row: macroblocks whose vectors are final and stored (what the search wave of the row below waits for)
 *   rowbits     one bit buffer per macroblock row (MSB-first 32-bit words)
 *   mbrec       per macroblock {mv[0], type, used-candidates} for the mv_clusters validation
 *   arena       finished slice RBSPs, written by the frame's finalizer workgroup
 */
#ifndef H264E_DEV_H
#define H264E_DEV_H
#include <stdint.h>

#define H264E_MV_NA 0x8000
#define H264E_ORDER_PAD 0xffffffffu     /* an entry of the dispatch order that is nobody's (padding of a banded order, h264e_pool.h build_order): the workgroup exits at once */
#define H264E_MAX_SLICES 16             /* row-band slices per frame (== H264E_HIP_MAX_SLICES) */
#define H264E_ROW_BYTES_PER_MB 2048     /* capacity of a row bit buffer, per macroblock of the row */

/* Temporal wavefront (a P frame starts while its reference frame is still being encoded): macroblock (x, row) needs the
 * reference window x*16-24 .. x*16+39 by row*16-24 .. row*16+39 reconstructed AND deblocked.  Its last sample lies in
 * macroblock (x+2, row+2) at local (7,7): outside the 3 right columns / bottom rows that the neighbours to the right
 * and below still filter, so it is final once row+2 of the reference frame has published x+3 macroblocks. */
#define H264E_DEP_ROWS 2
#define H264E_DEP_COLS 3
#define H264E_FRAME_LAG (2*H264E_DEP_ROWS + H264E_DEP_COLS)    /* macroblock steps between consecutive frames' starts */
/* Narrow mode: only the part of the window that ends at local sample (12,11) of reference macroblock (x+1, row+1) counts as
 * valid (53 columns x 52 rows: 24 samples of reach to the left / up, 12 / 11 to the right / down); those samples are final
 * once row+1 of the reference frame has published x+2 macroblocks (the pending bottom lines 12..15 are not needed), so
 * consecutive frames start 4 steps apart.  Vectors that reach further right / down take the HBM path with its dynamic
 * wait; the host falls back to the wide geometry when a clip does that often (h264e_host.c). */
#define H264E_NARROW_VW 53
#define H264E_NARROW_VH 52
#define H264E_NARROW_DEP_ROWS 1
#define H264E_NARROW_DEP_COLS 2
#define H264E_NARROW_FRAME_LAG (2*H264E_NARROW_DEP_ROWS + H264E_NARROW_DEP_COLS)

typedef int32_t mv32;                   /* packed (y << 16) | (x & 0xffff), quarter-pel */

typedef struct
{
    int width, height;                  /* picture size */
    int W, H;                           /* coded size (multiples of 16) */
    int nmbx, nmby, nmb, cropping;
    int lim_x0, lim_y0, lim_x1, lim_y1; /* mv_limit, h264-lab.h:6322-6324 */
    int row_words;                      /* 32-bit words per row bit buffer */
    unsigned spin_limit;                /* bound of every in-kernel wait (polls with s_sleep between them); expiry = reported failure */
    int test_stall_row;                 /* fault injection (tests): this macroblock row of job 0 exits without ever publishing; -1 = off */
    int fz_wait_all;                    /* A/B measurements (H264E_FZ_WAIT_ALL=1): the finalizer waits for ALL rows of its frame before it walks / splices any */
} h264e_geom_t;

typedef struct
{
    uint8_t pix[32];                    /* UNFILTERED bottom line: 16 Y, 8 U, 8 V (intra prediction of the row below) */
    mv32 mv[4];                         /* bottom row of 4x4 motion vectors, H264E_MV_NA for intra */
    uint8_t nnz[8];                     /* CAVLC contexts: 4 Y, 2 U, 2 V */
    int8_t i4[4];                       /* intra 4x4 modes of the bottom blocks */
    uint8_t df_nz;                      /* coded-coefficient flags of the bottom 4x4 blocks (deblock strength) */
    int8_t type;
    uint8_t qp;
    uint8_t pad;
} h264e_mbbottom_t;

/* The bottom 4 luma / 2 chroma lines of a macroblock are not final until the row below has filtered the edge between
 * them: they wait here (deblocked by their own macroblock and by its right neighbour) and the row below writes the final
 * samples into the picture.  The picture therefore only ever holds final samples of rows that are complete, which is
 * what lets a frame be encoded again from macroblock row `first_row` with everything above it kept. */
typedef struct
{
    uint8_t y[64];                      /* luma rows 12..15 */
    uint8_t c[2][16];                   /* chroma rows 6..7 of U, V */
} h264e_mbpend_t;

typedef struct
{
    uint32_t nbits;                     /* bits in the row buffer */
    int32_t lead_skips;                 /* skipped macroblocks before the first coded one (== nmbx when none is coded) */
    int32_t trail_skips;                /* skipped macroblocks after the last coded one */
    int32_t overflow;
} h264e_rowmeta_t;

typedef struct
{
    mv32 mv0;
    int8_t type;
    uint8_t used_cand;                  /* the macroblock consumed the mv_clusters start candidates */
    uint8_t pad[2];
} h264e_mbrec_t;

typedef struct
{
    uint32_t offset;                    /* byte offset of the frame's first slice RBSP in the chain's arena; slice k follows at the
                                           sum of the 16-byte-rounded sizes of the slices before it */
    uint32_t nbytes;                    /* span of all slices (last one not rounded) */
    int32_t nslices;
    uint32_t slice_nbytes[H264E_MAX_SLICES];
    int32_t all_skipped;
    int32_t clusters_moved;             /* some macroblock's update would change the speculated mv_clusters state */
    int32_t overflow;
    int32_t far_reads;                  /* reference accesses outside the valid window (HBM path) */
    int32_t pad[2];
} h264e_frameout_t;

/* Verdict of a frame's exact mv_clusters walk, done by its finalizer workgroup on the device (stream mode, walk_on_device): the
 * next frame's finalizer starts its own walk from state_out once `flag` carries the launch id. */
#define H264E_WALK_OK 1
#define H264E_WALK_BAD 2                 /* a macroblock consumed rounded candidates that differ from the exact ones: first_bad */
#define H264E_WALK_VOID 3                /* a frame before it was BAD / aborted: nothing to say about this one */
typedef struct
{
    mv32 state_out[2];                  /* OK: exact state behind the frame; BAD: the walk's end state (prediction for the frames behind) */
    int32_t status, first_bad;
    int32_t pad[3];
    int32_t flag;                       /* written last: launch id */
} h264e_walkrec_t;

/* per-job result record in HOST (pinned, device-mapped) memory: lets the host consume frames while the launch runs */
typedef struct
{
    uint32_t nbytes;
    int32_t all_skipped, clusters_moved, overflow;
    int32_t far_reads;
    int32_t nslices;
    uint32_t slice_nbytes[H264E_MAX_SLICES];
    int32_t in_device;                  /* the NALs did not fit the host-mapped mirror: fetch them from the slot's device NAL arena */
    int32_t walk_status, first_bad;     /* device-side mv_clusters validation (0 = not done on the device) */
    mv32 state_out[2];
    int32_t done;                       /* written last: launch id when the job's results are complete, -launch id when it was aborted */
} h264e_hostdone_t;

typedef struct
{
    uint8_t *rec[2][3];
    h264e_mbbottom_t *bottom;
    h264e_mbpend_t *pend;               /* [nmb] */
    int *progress;                      /* [2*nmby]: rows' progress counters, then their `decided` counters */
    uint32_t *rowbits;
    h264e_rowmeta_t *rowmeta;
    h264e_mbrec_t *mbrec;               /* [frame slots][nmb] */
    uint8_t *arena;
    uint32_t arena_cap;
    uint8_t *nal_arena;                 /* the frame's slices as finished Annex-B NALs (start code + escapes), 16-byte aligned behind each other */
    uint32_t nal_cap;
    uint32_t *cursor;
    h264e_frameout_t *fout;             /* [frame slots] */
    int *far_reads;                     /* counter of the frame being encoded (rows add, the finalizer reads and clears) */
    unsigned long long *prof;           /* [32] phase cycle sums, written only by the -DH264E_STAMPS diagnostic build */
} h264e_chain_dev_t;

typedef struct
{
    const uint8_t *in[3];
    int in_stride[3];
    int active;
    int slice_type;                     /* 0 = P, 2 = I */
    int qp;
    int speed;
    int no_deblock;
    int chain;                          /* which chain's row/record/result buffers this job uses ... */
    const h264e_chain_dev_t *chain_desc; /* ... = its descriptor in device memory (the jobs of one launch may belong to different pools: launch groups) */
    int *errflag;                       /* device word of the job's pool: set when a bounded wait expires */
    int *stepflags;                     /* device [2]: {clusters_moved, overflow} of this job (plain mode) */
    const uint8_t *ref[3];              /* reference picture (coded size, stride = width); unused for I slices */
    uint8_t *dec[3];                    /* picture being built */
    int arena_reset;                    /* the result goes to the start of the chain's arena (one result per chain slot) */
    const int *dep_progress;            /* progress counters of the job that builds `ref` in the SAME launch, or NULL when ref is complete */
    int frame_slot;
    int first_row;                      /* macroblock rows above it are kept from the previous encode of this frame */
    int narrow;                         /* reference-window geometry the launch was made for (all jobs of a launch agree) */
    int hdr_nal;                        /* NAL header byte; the kernel appends ue(first_mb_in_slice), then ... */
    int hdr_nbits;                      /* ... hdr_nbits tail bits of the slice header, right-aligned in hdr_bits */
    uint64_t hdr_bits;
    int nslices;                        /* >= 1 */
    int16_t slice_row[H264E_MAX_SLICES + 1];    /* first macroblock row of every slice; slice_row[nslices] = nmby */
    mv32 clusters[2];                   /* speculated mv_clusters state for every macroblock of the frame ... */
    const mv32 *clusters_per_mb;        /* ... or, when not NULL, an exact per-macroblock trajectory [nmb][2] */
    uint16_t qdat[2][42];               /* quantizer tables (h264-lab.h:5839-5912), built by the host */
    int launch_id;                      /* > 0, unique per submit */
    const int *abort_word;              /* device memory, or NULL: the job stops once *abort_word == launch_id (raised by a finalizer whose walk
                                           failed, or by the host) */
    int walk_on_device;                 /* the finalizer validates the mv_clusters speculation itself */
    int walk_quiet;                     /* a failed walk of this job does not raise the launch's abort word (a leaf nobody depends on) */
    mv32 exact_state[2];                /* ... from this exact state when walk_prev is NULL (first job of the launch) */
    const h264e_walkrec_t *walk_prev;   /* ... else from the verdict of the job before it in stream order (same launch) */
    h264e_walkrec_t *walk_out;          /* this job's verdict */
    mv32 *traj_out;                     /* [nmb][2] the exact state in front of every macroblock, as walked (input of a re-encode) */
    h264e_hostdone_t *host_done;        /* host-mapped result record, or NULL */
    uint8_t *host_rbsp;                 /* host-mapped copy of the RBSP (capacity host_rbsp_cap), or NULL */
    uint32_t host_rbsp_cap;
    h264e_mbrec_t *host_mbrec;          /* host-mapped copy of the macroblock records [nmb], or NULL */
    unsigned long long *mb_counter;     /* device word of the job's pool: every row adds the macroblocks it reconstructed when it ends or stops (bookkeeping:
                                           processed vs delivered macroblocks, H264E_clip_stats_t.processed_mbs) */
} h264e_frame_task_t;

#endif
