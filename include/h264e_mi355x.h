/*
 * h264e_mi355x.h -- the reference's public encoder API, served by the MI355X HIP path.
 *
 * Drop-in boundary: the four entry points and three structs below are what an application written
 * against /root/reference/src/h264-lab.h:1-318 links to; layouts are ABI (x86-64 LP64, built with the
 * reference's default H264E_SVC_API=1, H264E_MAX_THREADS=0: create 56 B, run 48 B, io_yuv 40 B).
 *
 *   H264E_sizeof          replaces h264-lab.h:6868-6892   (same sizes as the reference reports)
 *   H264E_init            replaces h264-lab.h:6375-6407
 *   H264E_encode          replaces h264-lab.h:6654-6861   (macroblock loop runs as HIP kernels, include/h264e_hip.h)
 *   H264E_set_vbv_state   replaces h264-lab.h:6898-6913
 *
 * Scope: AVC baseline, key and P frames, one reference frame, one slice per frame or N
 * row-band slices (H264E_set_slices / H264E_clip_param_t.slices: the reference's H264E_MAX_THREADS build),
 * constant QP or frame-level rate control.  Long-term reference frame types, SVC layers, the temporal
 * denoiser, MB-level rate control and NALU-size slicing answer H264E_STATUS_BAD_PARAMETER at init /
 * H264E_STATUS_BAD_FRAME_TYPE at encode instead of silently producing a different stream.
 * The encoder needs a HIP device: without one H264E_init fails (H264E_STATUS_BAD_ARGUMENT) -- there is
 * no CPU fallback.
 */
#ifndef H264E_MI355X_H
#define H264E_MI355X_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* h264-lab.h:25-34 */
#define H264E_STATUS_SUCCESS                0
#define H264E_STATUS_BAD_ARGUMENT           1
#define H264E_STATUS_BAD_PARAMETER          2
#define H264E_STATUS_BAD_FRAME_TYPE         3
#define H264E_STATUS_SIZE_NOT_MULTIPLE_16   4
#define H264E_STATUS_SIZE_NOT_MULTIPLE_2    5
#define H264E_STATUS_BAD_LUMA_ALIGN         6
#define H264E_STATUS_BAD_LUMA_STRIDE        7
#define H264E_STATUS_BAD_CHROMA_ALIGN       8
#define H264E_STATUS_BAD_CHROMA_STRIDE      9

/* h264-lab.h:63-70 */
#define H264E_FRAME_TYPE_DEFAULT    0
#define H264E_FRAME_TYPE_KEY        6
#define H264E_FRAME_TYPE_I          5
#define H264E_FRAME_TYPE_GOLDEN     4
#define H264E_FRAME_TYPE_RECOVERY   3
#define H264E_FRAME_TYPE_P          2
#define H264E_FRAME_TYPE_DROPPABLE  1
#define H264E_FRAME_TYPE_CUSTOM     99

/* h264-lab.h:83-172 */
typedef struct H264E_create_param_tag
{
    int width;
    int height;
    int gop;                                /* key frame period; 0 = only the first frame; 1 = all intra */
    int vbv_size_bytes;                     /* selects the SPS level; VBV model for rate control */
    int vbv_overflow_empty_frame_flag;
    int vbv_underflow_stuffing_flag;
    int fine_rate_control_flag;             /* MB-level rate control: not supported (must be 0) */
    int const_input_flag;                   /* 0: the reconstruction is written back into the input planes */
    int max_long_term_reference_frames;     /* must be 0 */
    int enableNEON;                         /* ignored */
    int temporal_denoise_flag;              /* must be 0 */
    int sps_id;
    int num_layers;                         /* SVC: must be 0 or 1 */
    int inter_layer_pred_flag;
} H264E_create_param_t;

/* h264-lab.h:177-226 */
typedef struct H264E_run_param_tag
{
    int encode_speed;
    int frame_type;
    int long_term_idx_use;
    int long_term_idx_update;
    int desired_frame_bytes;
    int qp_min;
    int qp_max;
    int desired_nalu_bytes;                 /* must be 0: byte-budget slicing (h264-lab.h:6418-6430) is refused; row-band slices: H264E_set_slices() */
    void (*nalu_callback)(const unsigned char *nalu_data, int sizeof_nalu_data, void *token);
    void *nalu_callback_token;
} H264E_run_param_t;

/* h264-lab.h:231-237 */
typedef struct H264E_io_yuv_tag
{
    unsigned char *yuv[3];
    int stride[3];
} H264E_io_yuv_t;

typedef struct H264E_persist_tag H264E_persist_t;
typedef struct H264E_scratch_tag H264E_scratch_t;

int  H264E_sizeof(const H264E_create_param_t *param, int *sizeof_persist, int *sizeof_scratch);
int  H264E_init(H264E_persist_t *enc, const H264E_create_param_t *param);
int  H264E_encode(H264E_persist_t *enc, H264E_scratch_t *scratch, const H264E_run_param_t *run_param,
                  H264E_io_yuv_t *frame, unsigned char **coded_data, int *sizeof_coded_data);
void H264E_set_vbv_state(H264E_persist_t *enc, int vbv_size_bytes, int vbv_fullness_bytes);

/* ---- extensions (not in the reference) ---------------------------------------------------------------- */

/* The reference API has no destructor: device resources of an encoder are released when the
 * same persist blob is initialised again, at process exit, or explicitly here. */
void H264E_close(H264E_persist_t *enc);
/* Row-band slices per frame for the drop-in encoder: the reference selects this through H264E_create_param_t.max_threads,
 * a field that only exists when it is compiled with -DH264E_MAX_THREADS=N (h264-lab.h:142-170, :6511-6574) and that changes
 * the struct ABI; this library keeps the default ABI and takes the number here instead.  0 / 1 = one slice per frame,
 * 2..16 = N slices (deblocking idc 2, contexts / availability / mv_clusters restarted per slice).  Call after H264E_init. */
int  H264E_set_slices(H264E_persist_t *enc, int nslices);
/* Select the HIP device used by subsequent H264E_init calls of this process (default 0 / $H264E_DEVICE). */
void H264E_set_device(int device);
int  H264E_device_count(void);
/* Last device-side error text (empty when none). */
const char *H264E_last_error(void);

/* Whole-clip streaming encode: consecutive frames run as a temporal wavefront inside one
 * kernel launch, finished frames are validated and NAL-assembled by the host while the launch runs (DESIGN.md
 * sections 4-5).  Bit-identical to feeding the frames one by one to H264E_encode.  With rate control (kbps) a frame's QP depends on the size
 * of the frame before it: the frames behind the first one of a launch run on a speculated QP and are stopped when the exact
 * controller disagrees. */
typedef struct
{
    int width, height, gop, qp, speed;
    int vbv_size_bytes;                     /* SPS level only */
    int device;
    int max_chains;                         /* cap of the slot ring (frame f lives in slot f % K; K - 1 = most frames one launch can hold); 0 = sized by the
                                               memory budget: 622 slots at 1080p, 162 at 4K, 40 at 8K, at most 1024 */
    int first_idr_pic_id_state;             /* enc->next_idr_pic_id before the first frame (0 for a fresh stream) */
    int32_t mv_clusters_in[2];              /* enc->mv_clusters before the first frame (0,0 for a fresh stream) */
    int slices;                             /* row-band slices per frame: 0 / 1 = one; N = the reference's H264E_MAX_THREADS build with --threads N */
    int kbps;                               /* 0 = constant QP `qp`; > 0 = frame-level rate control as encode_app --kbps (a few frames per launch then, on a speculated QP that is validated against the exact controller) */
    int resident_frames;                    /* input frames kept in HBM (a ring, frame f in slot f % resident_frames); 0 = the whole clip */
    int keep_records;                       /* keep what every frame consumed, for H264E_clip_revalidate (GOP shards of one stream) */
} H264E_clip_param_t;

typedef struct
{
    double upload_ms, encode_ms, readback_ms, assemble_ms;      /* host wall clock of the phases */
    double mb_kernel_ms, splice_kernel_ms;                      /* HIP-event time inside the kernel launches (when profiled); the splice runs inside the macroblock kernel: second value ~0 */
    int kernel_launches;
    int chains, rounds, reencoded_gops;                         /* frames in flight per launch, launches, relaunches after a mis-speculated mv_clusters state */
    int32_t mv_clusters_out[2];
    int next_idr_pic_id_state;
    int first_frame, frames;                                    /* the frames this call encoded: [first_frame, first_frame + frames) */
    int spin_relaunches;                                        /* launches repeated because a bounded in-kernel wait expired (workgroups starved of wave slots: nothing wrong was returned) */
    /* event bookkeeping of this call (round-3 VERDICT item 4): what the kernel worked on against what was delivered */
    int relaunches_timed;                                       /* launches that followed a stopped one (mis-speculated mv_clusters state, rate-control miss) and delivered a frame */
    long long processed_mbs;                                    /* macroblocks reconstructed by the kernel, including those of frames that were thrown away */
    long long delivered_mbs;                                    /* macroblocks of the frames that were accepted = frames x macroblocks per frame */
    double first_frame_ms_after_relaunch;                       /* SUM over relaunches_timed launches of: submit -> first accepted frame (host wall clock) */
} H264E_clip_stats_t;

/* sizeof of the extension structs as THIS build sees them (0: H264E_clip_param_t, 1: H264E_clip_stats_t): lets a binding in another language check its mirror */
int  H264E_struct_size(int which);

typedef struct H264E_clip_tag H264E_clip_t;
/* Create a clip encoder for a stream of at most nframes frames. */
int  H264E_clip_open(H264E_clip_t **clip, const H264E_clip_param_t *par, int nframes);
/* Input: packed I420 frames [first, first + nframes) of the stream from host memory, or the synth_v1 test clip generated in HBM.
 * With a bounded input ring (resident_frames) a frame can be uploaded once frame (f - resident_frames) has been encoded. */
int  H264E_clip_upload(H264E_clip_t *clip, int first, int nframes, const uint8_t *i420);
int  H264E_clip_generate_synth(H264E_clip_t *clip, int first, int nframes, int t0, uint32_t seed);
/* the same from pinned host memory on the copy engine, overlapping with a running encode; H264E_clip_upload_wait() completes it */
void *H264E_clip_host_alloc(size_t bytes);
void  H264E_clip_host_free(void *p);
int  H264E_clip_upload_async(H264E_clip_t *clip, int first, int nframes, const uint8_t *pinned_i420);
int  H264E_clip_upload_wait(H264E_clip_t *clip);
int  H264E_clip_upload_poll(H264E_clip_t *clip);              /* 1 = landed (frames count as uploaded), 0 = still copying */
/* where the stream stands: next frame to encode, frames uploaded so far */
void H264E_clip_position(const H264E_clip_t *clip, int *next_frame, int *uploaded_frames);
/* called (from the calling thread) while H264E_clip_encode waits for the GPU: a file-fed application issues and completes its
 * uploads from here, so that frames keep arriving while earlier ones are being encoded */
void H264E_clip_set_idle_hook(H264E_clip_t *clip, void (*hook)(void *token), void *token);
/* Encode the frames uploaded so far and not yet encoded -- or as many of them as fit `out` -- and append their Annex-B bytes;
 * stats->first_frame / frames say which; frame_bytes[i] is the size of frame first_frame + i.  The stream continues with the
 * next call.  profile != 0 adds per-kernel HIP-event timing. */
int  H264E_clip_encode(H264E_clip_t *clip, uint8_t *out, size_t cap, size_t *out_bytes, int *frame_bytes /* [frames of this call] or NULL */,
                       int profile, H264E_clip_stats_t *stats);
/* Several independent clips of the same picture size on ONE device at the same time: each clip is encoded exactly like
 * H264E_clip_encode does (same bytes), by a host thread of its own, and the device layer merges the clips' kernel launches so that one
 * stream's pipeline drains (after a mis-speculated mv_clusters state) are filled by the other streams' frames.  out[i] / cap[i] /
 * out_bytes[i] / frame_bytes[i] / stats[i] belong to clips[i] (frame_bytes and stats may be NULL).  At most 8 clips; all on the same
 * device; large pictures allow fewer (the group refuses a clip it cannot hold safely). */
int  H264E_clip_encode_multi(H264E_clip_t **clips, int nclips, uint8_t **out, const size_t *cap, size_t *out_bytes, int **frame_bytes, H264E_clip_stats_t *stats);
/* back to frame 0 with the stream state of H264E_clip_open (the uploaded frames stay): encode the clip again */
void H264E_clip_rewind(H264E_clip_t *clip);
/* resident input frames back to host memory (whole-clip residency only; bench.py's PCIe-inclusive measurement) */
int  H264E_clip_download(H264E_clip_t *clip, int first, int nframes, uint8_t *i420);
/* reconstruction (coded size, packed I420) of one of the last frames encoded (its picture slot must not have been reused: the last
 * ring - 1 frames in constant-QP mode; with rate control only the frames accepted since the last launch began -- the launches' hedge
 * leaves are encoded over the pictures of older frames).  -1 outside that window. */
int  H264E_clip_read_recon(H264E_clip_t *clip, int frame, uint8_t *dst);
/* per encoded frame [3] sums of squared differences input vs reconstruction (Y, U, V), computed on the device: encode_app --psnr */
void H264E_clip_set_ssd_output(H264E_clip_t *clip, uint64_t *ssd);
/* GOP shards of ONE stream (one clip encoder per shard / GPU, the shard starting at a key frame with first_idr_pic_id_state =
 * its GOP index & 1 and a SPECULATED mv_clusters_in): once the exact state in front of the shard is known, revalidate walks the
 * kept records (keep_records) and either confirms the shard (*restart_frame = -1, end_state = exact state behind it) or names the
 * GOP from which it has to be encoded again (restart_frame, restart_state); H264E_clip_restart rewinds the encoder to that point. */
int  H264E_clip_revalidate(H264E_clip_t *clip, const int32_t exact_in[2], int *restart_frame, int32_t restart_state[2], int32_t end_state[2]);
int  H264E_clip_restart(H264E_clip_t *clip, int frame, const int32_t state[2]);
/* kept records of an encoded frame (keep_records): per macroblock {int32 mv[0] packed (y << 16) | (x & 0xffff); int8 type: -1 skip,
 * 0..3 inter partitioning, 5 I4x4, 6 I16x16; uint8 used-cluster-candidates; 2 pad bytes} -- a per-macroblock trace for debugging */
int  H264E_clip_read_records(H264E_clip_t *clip, int frame, void *dst /* macroblocks x 8 bytes */);
void H264E_clip_close(H264E_clip_t *clip);
/* diagnostic: per-phase sums [48] of a -DH264E_STAMPS kernel build since the last call (zeros in the product): [0..31] the rows' phases in cycles, [32..47] the finalizers' in 10 ns ticks */
int  H264E_clip_stamps(H264E_clip_t *clip, unsigned long long *dst);

#ifdef __cplusplus
}
#endif
#endif
