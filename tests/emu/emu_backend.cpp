/*
 * emu_backend.cpp -- TEST ONLY: the lane-loop emulation of the macroblock kernels (h264-lab_amd/csrc/wave.h, -DH264E_EMU) behind the
 * product's own host layer.  The kernel HEADERS (enc_*.h) are the product's; a "launch" here runs the jobs of the launch one after
 * the other, every macroblock row as one call sequence row_begin / row_step ... / row_end, then the job's finalizer -- the order the
 * GPU's dependency counters would allow anyway.  It checks kernel LOGIC without a GPU; address spaces, the memory model and the
 * wave pipeline's hand-offs are what the -m gpu tests are for.  Never linked into libh264e_mi355x.so.
 */
#ifndef H264E_EMU
#error "the emulation is built with -DH264E_EMU (tests/emu/Makefile)"
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "emu_hip.h"

/* ---- the registry of "device" memory and the check behind the kernel sources' global-memory accessors (wave.h EMU_GLOBAL) */
static pthread_mutex_t g_blk_mu = PTHREAD_MUTEX_INITIALIZER;
static struct { const char *lo, *hi; } g_blk[4096];
static int g_nblk;
void emu_register_device_block(void *p, size_t n)
{
    pthread_mutex_lock(&g_blk_mu);
    if (g_nblk < 4096) { g_blk[g_nblk].lo = (const char *)p; g_blk[g_nblk].hi = (const char *)p + n; g_nblk++; }
    pthread_mutex_unlock(&g_blk_mu);
}
void emu_unregister_device_block(void *p)
{
    pthread_mutex_lock(&g_blk_mu);
    for (int i = 0; i < g_nblk; i++) if (g_blk[i].lo == (const char *)p) { g_blk[i] = g_blk[--g_nblk]; break; }
    pthread_mutex_unlock(&g_blk_mu);
}
extern "C" void emu_check_global(const void *p, size_t n, const char *file, int line)
{
    int ok = 0;
    pthread_mutex_lock(&g_blk_mu);
    for (int i = 0; i < g_nblk && !ok; i++) ok = (const char *)p >= g_blk[i].lo && (const char *)p + n <= g_blk[i].hi;
    pthread_mutex_unlock(&g_blk_mu);
    if (!ok)
    {
        fprintf(stderr, "%s:%d: global-memory access to %p (%zu bytes), which is not device memory: a stack / register / LDS address was cast to a global pointer\n", file, line, p, n);
        abort();
    }
}

#include "../../h264-lab_amd/csrc/enc_row.h"
#include "../../h264-lab_amd/csrc/enc_selftest.h"
#include "../../include/h264e_hip.h"

/* variant 0 = a launch of intra frames only (the kernel variant without inter code), like h264e_kernels.hip bk_launch_mb */
static void bk_launch_mb(const h264e_geom_t &G, int narrow, int variant, int njobs, unsigned nblocks, const h264e_frame_task_t *tasks, const uint32_t *order, hipStream_t)
{
    (void)order; (void)nblocks;
    for (int job = 0; job < njobs; job++)
    {
        const h264e_frame_task_t &T = tasks[job];
        if (!T.active) continue;
        const ChainG C = chain_view(*T.chain_desc);
        for (int row = T.first_row; row < G.nmby; row++)
        {
            RowLds *L = (RowLds *)calloc(1, sizeof(RowLds));
            row_begin(*L, G, C, T, row);
            int row0 = 0, row1 = G.nmby;
            for (int k = 0; k < T.nslices; k++)
                if (row >= T.slice_row[k] && row < T.slice_row[k + 1]) { row0 = T.slice_row[k]; row1 = T.slice_row[k + 1]; }
            const RowTask RT = rowtask_load(T);
            for (int x = 0; x < G.nmbx; x++)
            {
                if (variant == 0) { row_prefetch<GEOM_INTRA>(*L, G, RT, row, x); row_step<GEOM_INTRA>(*L, G, C, RT, row, x, row0, row1); }
                else if (narrow) { row_prefetch<GEOM_NARROW>(*L, G, RT, row, x); row_step<GEOM_NARROW>(*L, G, C, RT, row, x, row0, row1); }
                else { row_prefetch<GEOM_WIDE>(*L, G, RT, row, x); row_step<GEOM_WIDE>(*L, G, C, RT, row, x, row0, row1); }
            }
            row_end(*L, G, C, row);
            free(L);
        }
        /* the job's finalizer (h264e_kernels.hip, workgroup `nmby`): the same row-by-row walk + splice; every row is complete here */
        int wstatus = 0, first_bad = -1;
        mv32 ws[2] = { T.exact_state[0], T.exact_state[1] };
        if (T.walk_on_device && T.walk_prev)
        {
            if (T.walk_prev->flag != T.launch_id || T.walk_prev->status != H264E_WALK_OK) wstatus = H264E_WALK_VOID;
            else { ws[0] = T.walk_prev->state_out[0]; ws[1] = T.walk_prev->state_out[1]; }
        }
        JobWalk W;
        W.begin(ws);
        SpliceOut so;
        const bool walking = T.walk_on_device && !wstatus;
        const h264e_mbrec_t *rec = C.mbrec + (size_t)T.frame_slot*G.nmb;
        if (!wstatus)
            (void)splice_frame(G, C, T, so, [&](int, int r) -> int { if (walking) W.rows(G, T, rec, T.traj_out, r); return 0; });
        if (T.walk_on_device)
        {
            if (!wstatus)
            {
                W.end(T);
                first_bad = W.first_bad;
                ws[0] = W.s[0]; ws[1] = W.s[1];
                wstatus = first_bad >= 0 ? H264E_WALK_BAD : H264E_WALK_OK;
            }
            if (T.walk_out) { T.walk_out->state_out[0] = ws[0]; T.walk_out->state_out[1] = ws[1]; T.walk_out->status = wstatus; T.walk_out->first_bad = first_bad; T.walk_out->flag = T.launch_id; }
            if (wstatus != H264E_WALK_OK)
            {
                if (T.host_done)
                {
                    T.host_done->walk_status = wstatus; T.host_done->first_bad = first_bad; T.host_done->state_out[0] = ws[0]; T.host_done->state_out[1] = ws[1];
                    T.host_done->done = -T.launch_id;
                }
                continue;
            }
        }
        finalize_commit(G, C, T, so, T.stepflags);
        if (T.host_done)
        {
            uint32_t nal_bytes[H264E_MAX_SLICES], nal_total = 0;
            int exp_overflow = 0, in_device = 0;
            export_frame(G, C, T, nal_bytes, nal_total, exp_overflow, in_device);
            const h264e_frameout_t &F = C.fout[T.frame_slot];
            T.host_done->nbytes = nal_total; T.host_done->all_skipped = F.all_skipped;
            T.host_done->nslices = F.nslices; T.host_done->in_device = in_device;
            for (int k = 0; k < H264E_MAX_SLICES; k++) T.host_done->slice_nbytes[k] = nal_bytes[k];
            T.host_done->clusters_moved = F.clusters_moved; T.host_done->overflow = F.overflow | exp_overflow; T.host_done->far_reads = F.far_reads;
            T.host_done->walk_status = wstatus; T.host_done->first_bad = first_bad; T.host_done->state_out[0] = ws[0]; T.host_done->state_out[1] = ws[1];
            T.host_done->done = T.launch_id;
        }
    }
}

static void bk_launch_synth(uint8_t *dst, int w, int h, int t, uint32_t seed, hipStream_t)
{
    const int n = w*h*3/2;
    for (int i = 0; i < n; i++) dst[i] = sv_sample(w, h, t, seed, i);
}

static void bk_launch_ssd(int n, const uint8_t *clip, size_t frame_bytes, int width, int height, int in0, int in_mod, const h264e_chain_dev_t *chains, int pic0, int pic_mod, int W,
                          unsigned long long *out, hipStream_t)
{
    for (int i = 0; i < n; i++)
        for (int pl = 0; pl < 3; pl++)
        {
            const int w = width >> (pl ? 1 : 0), h = height >> (pl ? 1 : 0), ps = W >> (pl ? 1 : 0);
            const uint8_t *a = clip + frame_bytes*(size_t)((in0 + i) % in_mod) + (pl ? (size_t)width*height + (pl == 2 ? (size_t)(width/2)*(height/2) : 0) : 0);
            const uint8_t *b = chains[(pic0 + i) % pic_mod].rec[0][pl];
            unsigned long long s = 0;
            for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) { const int d = (int)a[(size_t)y*w + x] - (int)b[(size_t)y*ps + x]; s += (unsigned long long)(d*d); }
            out[3*i + pl] += s;
        }
}

static void bk_launch_nal_selftest(uint8_t *dst, uint32_t cap, const uint8_t *src, uint32_t n, uint32_t *out, hipStream_t)
{
    int overflow = 0;
    out[0] = nal_escape_copy(dst, cap, src, n, overflow);
    out[1] = (uint32_t)overflow;
}

static void bk_launch_stage_selftest(int stage, const uint8_t *in, const int *args, uint8_t *out, hipStream_t)
{
    StageLds *S = (StageLds *)calloc(1, sizeof(StageLds));
    RowLds *L = (RowLds *)calloc(1, sizeof(RowLds));
    if (S && L) stage_selftest(*S, *L, stage, in, args, out);
    free(S); free(L);
}

#include "../../h264-lab_amd/csrc/h264e_pool.h"
