/*
 * h264e_hip.h -- thin C ABI over the HIP kernels of the per-frame encode path (libh264e_mi355x.so).
 *
 * This is the device boundary the C host code (h264e_host.c: H264E_sizeof / H264E_init / H264E_encode,
 * encode_app) sits on.  It replaces, for the macroblock loop, the reference's internal call
 *     H264E_encode_one -> encode_slice -> mb_encode          (/root/reference/src/h264-lab.h:6477, :6409, :5724)
 * i.e. everything the reference reaches through its function table h264e_* (h264-lab.h:3274-3364).
 * Plain pointers and sizes only; no C++ or torch types.
 *
 * A POOL holds `nchains` slots of one picture geometry on one GPU (picture, row bit buffers, per-macroblock
 * records, result buffers).  One h264e_hip_submit() call launches the macroblock kernel once for up to nchains
 * jobs (one wavefront per macroblock row plus one finalizer wavefront per job that splices the slice):
 *  - plain mode: job c is the next frame of independent chain c (reference / reconstruction ping-pong);
 *  - stream mode: the jobs are consecutive frames of ONE stream, run as a temporal wavefront (a P frame starts
 *    while its reference frame is a few macroblock rows ahead); finished frames are
 *    exported to host-mapped memory and can be consumed with h264e_hip_stream_* while the launch runs.
 * Calls are asynchronous on the pool's stream until h264e_hip_sync().
 */
#ifndef H264E_HIP_H
#define H264E_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct h264e_hip_pool h264e_hip_pool_t;

/* one frame of one chain */
typedef struct
{
    int active;                 /* 0: chain idles in this step */
    int frame_index;            /* which resident input frame (see h264e_hip_upload_*) */
    int frame_slot;             /* where the result goes: 0 .. slots_per_chain-1 */
    int slice_type;             /* 0 = P, 2 = I   (h264-lab.h:3203-3204) */
    int qp;                     /* frame QP, 10..51 */
    int speed;                  /* H264E_run_param_t.encode_speed (h264-lab.h:181) */
    /* slice header (h264-lab.h:4182-4333) as a template: NAL header byte, then ue(first_mb_in_slice) -- written by the kernel
     * for every slice -- then `hdr_nbits` <= 56 tail bits (slice_type ... deblocking fields), right-aligned in hdr_bits */
    int hdr_nal;
    int hdr_nbits;
    uint64_t hdr_bits;
    /* row-band slices (h264-lab.h:6511-6574, the reference's H264E_MAX_THREADS build): 0 / 1 = one slice per frame; N > 1 =
     * N slices of consecutive macroblock rows, split like the reference (h264-lab.h:6530); <= H264E_HIP_MAX_SLICES */
    int nslices;
    int32_t mv_clusters[2];     /* speculated enc->mv_clusters for the whole frame (h264-lab.h:766) */
    const int32_t *mv_clusters_per_mb;  /* optional HOST array [nmb][2]: exact per-macroblock values (re-encode path) */
    uint16_t qdat[2][42];       /* quantizer tables of rc_set_qp (h264-lab.h:5839-5912) */
    /* temporal wavefront (stream_mode = 1): the tasks of one submit are consecutive frames of ONE stream, in order;
     * task i builds the picture of chain slot `slot` and references the picture of `ref_slot` (-1: none, I slice),
     * which is either complete or (ref_in_flight) being built by an earlier task of this same submit, a few rows ahead */
    int stream_mode, slot, ref_slot, ref_in_flight;
    /* stream mode: encode this frame again from macroblock row first_row; the rows above it (bits, records, picture) are
     * kept from the previous encode of the same frame in the same slot (0 = whole frame) */
    int first_row;
    /* stream mode, device walk: index + 1 of the task whose verdict this frame's walk starts from; 0 = the task in front of it */
    int walk_parent;
    /* ... and a frame whose failed validation concerns nobody but itself (a leaf): it reports the verdict but does not stop the launch */
    int walk_quiet;
    /* stream mode: the job's finalizer validates the mv_clusters speculation ON THE DEVICE (exact walk over the frame's records,
     * enc_row.h device_clusters_walk): task 0 of a submit starts from exact_state, task i from the verdict of task i-1; a mismatch
     * stops the launch (device abort word) and is reported in the result (walk_status, first_bad, state_out); the walked
     * per-macroblock trajectory stays on the device and becomes the candidates of a re-encode with traj_from_device */
    int walk_on_device;
    int32_t exact_state[2];
    int traj_from_device;
    /* stream mode: 1 = narrow valid window (53 x 52 samples, consecutive frames 4 macroblock steps apart), 0 = the whole
     * 64 x 64 window (7 steps apart); see h264e_dev.h.  Same bits either way. */
    int narrow_window;
} h264e_hip_task_t;

#define H264E_HIP_MAX_SLICES 16

typedef struct
{
    uint32_t nbytes;            /* bytes of the frame's slice RBSPs as exported: slice k starts at the sum of the 16-byte-rounded sizes before it */
    int nslices;
    uint32_t slice_nbytes[H264E_HIP_MAX_SLICES];   /* RBSP bytes of each slice NAL (header byte included, no start code, no escapes) */
    int all_skipped;            /* every macroblock was skipped (rc_frame_end's skip_flag, h264-lab.h:6596) */
    int clusters_moved;         /* the speculated mv_clusters state is not a fixed point of this frame */
    int overflow;               /* a bit buffer overflowed: the result is invalid */
    int far_reads;              /* reference accesses that left the valid window and took the HBM path (stream mode) */
    int in_device;              /* the NALs did not fit the host-mapped mirror: h264e_hip_stream_fetch_nals() gets them */
    int walk_status;            /* device-side validation: 0 not done, 1 ok, 2 this frame consumed wrong candidates (first_bad), 3 void (a frame before it failed) */
    int first_bad;
    int32_t state_out[2];       /* ok: exact mv_clusters behind the frame; bad: the walk's end state */
} h264e_hip_result_t;

typedef struct { int32_t mv0; int8_t type; uint8_t used_cand; uint8_t pad[2]; } h264e_hip_mbrec_t;

int  h264e_hip_device_count(void);
/* frames_resident: input frames kept in HBM; slots_per_chain: results kept per chain between reads */
int  h264e_hip_pool_create(h264e_hip_pool_t **pool, int device, int width, int height, int nchains,
                           int frames_resident, int slots_per_chain);
void h264e_hip_pool_destroy(h264e_hip_pool_t *pool);
/* packed I420 frames (width*height*3/2 bytes each) from host memory into resident slots first.. */
int  h264e_hip_upload_i420(h264e_hip_pool_t *pool, int first, int nframes, const uint8_t *host_i420);
/* the same from pinned host memory (h264e_hip_host_alloc) on the pool's copy stream, concurrently with kernels on the encode
 * stream; h264e_hip_upload_wait() blocks until every such copy has landed */
int  h264e_hip_upload_i420_async(h264e_hip_pool_t *pool, int first, int nframes, const uint8_t *pinned_i420);
int  h264e_hip_upload_wait(h264e_hip_pool_t *pool);
int  h264e_hip_upload_busy(h264e_hip_pool_t *pool);         /* 1 while such a copy is still in flight, 0 when all have landed, -1 when the copy stream failed */
void *h264e_hip_host_alloc(size_t bytes);
void h264e_hip_host_free(void *p);
/* one frame from three planes with arbitrary strides (the H264E_io_yuv_t of the drop-in API) */
int  h264e_hip_upload_planes(h264e_hip_pool_t *pool, int index, const uint8_t *const yuv[3], const int stride[3]);
/* fill resident frames [first, first+n) with the synth_v1 test clip ON THE DEVICE (bench input, already in HBM) */
int  h264e_hip_generate_synth(h264e_hip_pool_t *pool, int first, int nframes, int t0, uint32_t seed);
int  h264e_hip_submit(h264e_hip_pool_t *pool, const h264e_hip_task_t *tasks /* [nchains] */);
/* Launch groups: the submits of the member pools (same device, same picture size, one host thread each) are merged into ONE kernel
 * launch per round -- the streams' jobs interleaved in dispatch order, every job with its own pool's buffers / abort word / results --
 * so that independent streams fill each other's pipeline drains.  h264e_hip_submit of a member blocks until all members that are
 * still in the group have submitted (or left); h264e_hip_sync returns when the merged launch has drained. */
typedef struct h264e_hip_group h264e_hip_group_t;
int  h264e_hip_group_create(h264e_hip_group_t **group, int device);
int  h264e_hip_group_join(h264e_hip_group_t *group, h264e_hip_pool_t *pool);
void h264e_hip_group_leave(h264e_hip_group_t *group, h264e_hip_pool_t *pool);
void h264e_hip_group_destroy(h264e_hip_group_t *group);
int  h264e_hip_sync(h264e_hip_pool_t *pool);
/* {clusters_moved, overflow} of every chain for the LAST submitted step, in one copy (call after h264e_hip_sync) */
int  h264e_hip_step_flags(h264e_hip_pool_t *pool, int *flags /* [nchains][2] */);
/* streaming (pools with slots_per_chain == 1, tasks with stream_mode): every job has a finalizer inside the launch that
 * copies its RBSP and macroblock records to host-mapped memory and raises a done word, so the host can consume frame
 * after frame while later frames of the same launch are still running.
 * h264e_hip_stream_done: 0 not finished, 1 finished (res filled), 2 aborted. */
/* A pool owns its device's launch lock (process-wide, one persistent launch at a time per device) from its first submit until
 * h264e_hip_sync returns; h264e_hip_release gives it back after a failure in between (waits for the stream, reports nothing). */
void h264e_hip_release(h264e_hip_pool_t *pool);
int  h264e_hip_stream_done(h264e_hip_pool_t *pool, int slot, h264e_hip_result_t *res);
const uint8_t *h264e_hip_stream_rbsp(h264e_hip_pool_t *pool, int slot);
const h264e_hip_mbrec_t *h264e_hip_stream_mbrec(h264e_hip_pool_t *pool, int slot);
/* a per-macroblock trajectory of `slot` ([nmb][2]): consumed = 0 the one its last device walk produced (what a re-encode with
 * traj_from_device will consume), 1 the one its last encode consumed (traj_from_device) */
int  h264e_hip_stream_fetch_traj(h264e_hip_pool_t *pool, int slot, int consumed, int32_t *dst);
int  h264e_hip_stream_fetch_nals(h264e_hip_pool_t *pool, int slot, uint8_t *dst, uint32_t nbytes);
/* after the launch has drained: the reconstructed picture of slot `from` becomes the picture of slot `to` (a frame that was
 * encoded in a spare slot is moved to the slot its frame number owns) */
int  h264e_hip_stream_copy_picture(h264e_hip_pool_t *pool, int from, int to);
/* resident input frames back to the host (measurement helper) */
int  h264e_hip_download_i420(h264e_hip_pool_t *pool, int first, int nframes, uint8_t *host_i420);
int  h264e_hip_stream_abort(h264e_hip_pool_t *pool);
int  h264e_hip_busy(h264e_hip_pool_t *pool);
int  h264e_hip_result(h264e_hip_pool_t *pool, int chain, int slot, h264e_hip_result_t *res);
int  h264e_hip_read_rbsp(h264e_hip_pool_t *pool, int chain, int slot, uint8_t *dst, uint32_t cap);
/* all results of a chain in two copies: per-slot result + byte offset into arena_dst, which receives the used part of the arena */
int  h264e_hip_read_chain(h264e_hip_pool_t *pool, int chain, int nslots, h264e_hip_result_t *res, uint32_t *offsets,
                          uint8_t *arena_dst, uint32_t cap, uint32_t *used);
int  h264e_hip_read_mbrec(h264e_hip_pool_t *pool, int chain, int slot, h264e_hip_mbrec_t *dst /* [nmb] */);
/* records of slots 0..nslots-1 in one copy, dst[nslots][nmb] */
int  h264e_hip_read_mbrec_all(h264e_hip_pool_t *pool, int chain, int nslots, h264e_hip_mbrec_t *dst);
/* reconstructed picture of the chain's last frame, coded size, packed I420 */
int  h264e_hip_read_recon(h264e_hip_pool_t *pool, int chain, uint8_t *dst);
/* stream pools: the picture of chain slot `slot` (coded size, packed I420) */
int  h264e_hip_read_recon_slot(h264e_hip_pool_t *pool, int slot, uint8_t *dst);
/* sums of squared differences between resident input frames and stream pictures, one small kernel per call: frame i uses input
 * slot (in0 + i) % in_mod and picture slot (pic0 + i) % pic_mod; out = host [n][3] (Y, U, V), picture size width x height */
int  h264e_hip_ssd_frames(h264e_hip_pool_t *pool, int n, int in0, int in_mod, int pic0, int pic_mod, uint64_t *out);
/* forget the results of a chain (arena cursor back to 0); the reference picture is kept */
int  h264e_hip_reset_results(h264e_hip_pool_t *pool, int chain);
/* drop the chain's last submitted frame (result in `slot`): undo the reference/reconstruction swap and give its
 * arena space back, so the frame can be submitted again (re-encode path) */
int  h264e_hip_rewind_frame(h264e_hip_pool_t *pool, int chain, int slot);
/* kernel timing on the pool's stream (HIP events around every macroblock-kernel launch) */
/* macroblocks the pool's rows have reconstructed since the last reset, delivered or thrown away (call after h264e_hip_sync) */
int h264e_hip_mb_counter(h264e_hip_pool_t *pool, unsigned long long *count, int reset);
void h264e_hip_profile(h264e_hip_pool_t *pool, int enable);
int  h264e_hip_profile_read(h264e_hip_pool_t *pool, double *mb_kernel_ms, double *splice_kernel_ms, int *launches);
/* diagnostic: per-phase cycle sums of a -DH264E_STAMPS build of the kernels (all zero in the product build) */
int  h264e_hip_stamps_read(h264e_hip_pool_t *pool, unsigned long long *dst /* [32] */, int reset);
/* wall clock of a region on the pool's stream, by HIP events */
int  h264e_hip_timer_start(h264e_hip_pool_t *pool);
int  h264e_hip_timer_stop(h264e_hip_pool_t *pool, double *ms);
/* self-test hook: runs the device's NAL emulation-prevention pass (enc_row.h nal_escape_copy, one wavefront) on n payload bytes;
 * dst receives start code + escaped payload, *out_n its size.  Used by tests with adversarial inputs (real streams need an
 * escape about once per 4 MB). */
/* test hook: the dispatch order of a launch of `jobs` jobs as (job << 16 | row) entries, 0xffffffff = padding (banded orders: eight
 * equally long per-XCD queues, h264e_pool.h build_order); returns the number of entries, -1 on failure */
long h264e_hip_selftest_order(h264e_hip_pool_t *pool, int jobs, int narrow, int banded, uint32_t *out, size_t cap);
int  h264e_hip_selftest_nal_escape(h264e_hip_pool_t *pool, const uint8_t *src, uint32_t n, uint8_t *dst, uint32_t cap, uint32_t *out_n);
/* test hook: one wave-level stage of the macroblock pipeline on caller-supplied operands (h264e_kernels.hip stage_selftest lists
 * the stages and their operand layouts; tests/test_stages.py compares them with the reference's own functions) */
int  h264e_hip_selftest_stage(h264e_hip_pool_t *pool, int stage, const uint8_t *in, uint32_t nin, const int *args /* [24] */, uint8_t *out, uint32_t nout);
const char *h264e_hip_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
