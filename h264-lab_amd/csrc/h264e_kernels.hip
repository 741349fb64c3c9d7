/*
 * h264e_kernels.hip -- HIP kernels of the per-frame encode path and their C-ABI launcher (include/h264e_hip.h).
 *
 * Kernels (gfx950, wave64):
 *   h264e_mb_kernel      one 64-lane workgroup per (job, macroblock row) + one finalizer workgroup per job; rows of one
 *                        picture run as a wavefront behind each other: row r may encode macroblock x once row r-1 has
 *                        published x+2 macroblocks (left, top-left, top, top-right neighbours + in-loop deblocking order);
 *                        consecutive frames of a stream run as a temporal wavefront a few steps apart.  Hand-off: per-row
 *                        progress counters, payload stored write-through (sc1), relaxed polls, one acquire per hand-off
 *                        (cdna_hip_programming.md Guideline 16 R1).  The finalizer splices the row bit buffers into the
 *                        slice RBSP (there is no separate splice kernel) and exports the frame to host-mapped memory.
 *   h264e_synth_kernel   fills resident input frames with the synth_v1 clip (bench / test input in HBM).
 *
 * Built as the product with hipcc --offload-arch=gfx950.  The same file compiles with g++ -DH264E_EMU into
 * a lane-loop emulation that only tests/ use (see wave.h); the product library never contains that path.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "enc_row.h"
#include "../../include/h264e_hip.h"

#ifndef H264E_EMU
#include <hip/hip_runtime.h>
#endif

static thread_local char g_err[256];       /* per calling thread */
#define FAIL(...) do { snprintf(g_err, sizeof(g_err), __VA_ARGS__); return -1; } while (0)
extern "C" const char *h264e_hip_last_error(void) { return g_err; }

/* ------------------------------------------------------------------ device code */

#ifndef H264E_EMU


/* relaxed poll of one progress counter until it reaches `need`.  Returns 0 = reached, -1 = producer failed or the
 * bound expired, -2 = producer was aborted (negative counters are poison left behind by a row that stopped). */
DEV int poll_progress(const GLOBAL_AS int *flag, int need, int &seen, unsigned spin_limit)
{
    unsigned spins = 0;
    for (;;)
    {
        seen = uni(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));     /* uni: the loop control is scalar */
        if (seen >= need) return 0;
        if (seen < 0) return seen;
        if (++spins > spin_limit) return -1;
        __builtin_amdgcn_s_sleep(8);
    }
}

/* ---- hand-off words of the two-wave pipeline (LDS): release / acquire at workgroup scope, polled with s_sleep */
DEV int flag_get(const int *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
DEV void flag_set(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
#define LDS_SPIN_LIMIT (1u << 26)       /* the partner wave is resident: this bound only ever ends a wait when something is badly wrong */
/* 0 = *flag reached need; otherwise the stop code the partner wave (or this bound) left: -1 failure, -2 abort */
DEV int lds_wait(const int *flag, int need, int *stop)
{
    for (unsigned spins = 0;; spins++)
    {
        if (uni(flag_get(flag)) >= need) return 0;
        const int s = uni(flag_get(stop));
        if (s) return s;
        if (spins > LDS_SPIN_LIMIT) { flag_set(stop, -1); return -1; }
        __builtin_amdgcn_s_sleep(1);
    }
}
/* what the search wave tells the reconstruction wave while it is still searching macroblock need - 1 (enc_mb.h inter_choose) */
struct SearchSignals
{
    RowLds *L;
    int need;
    DEVM void noskip() const { flag_set(&L->f_noskip, need); }
    DEVM void bound(int u) const { L->early_bound = u; flag_set(&L->f_bound, need); }
};
/* the reconstruction wave's view of the inter decision of macroblock need - 1 (enc_row.h mb_intra_decide) */
struct InterFromSearchWave
{
    RowLds *L;
    int need;
    DEVM bool ready() const { return uni(flag_get(&L->f_inter)) >= need; }
    DEVM bool wait_ready() const { return lds_wait(&L->f_inter, need, &L->f_stop) == 0; }
    DEVM int early_bound() const { return uni(flag_get(&L->f_bound)) >= need ? uni(L->early_bound) : 0x7fffffff; }
    DEVM bool wait_noskip_or_ready() const
    {
        for (unsigned spins = 0;; spins++)
        {
            if (uni(flag_get(&L->f_inter)) >= need || uni(flag_get(&L->f_noskip)) >= need) return true;
            if (uni(flag_get(&L->f_stop))) return false;
            if (spins > LDS_SPIN_LIMIT) { flag_set(&L->f_stop, -1); return false; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
};

/*
 * Grid: njobs x (nmby + 1) workgroups of one wavefront -- or of two (WAVES = 2): the macroblock loop as a two-stage pipeline, one
 * wavefront searching macroblock x + 1 while the other reconstructs and writes x (enc_row.h).  Workgroup `row < nmby` encodes macroblock row
 * `row` of its job's frame; workgroup `nmby` is the job's finalizer: once every row has ended it splices the slice
 * (finalize_frame) and, in streaming use, exports the result to host-mapped memory and raises the job's done word,
 * so the host consumes frames while later frames of the same launch are still being encoded.
 * Every workgroup waits for workgroups with a lower index (rows above; rows of the reference frame's job inside the
 * static lag; rows of the own job for the finalizer) -- with ONE exception: a reference read that leaves the LDS window
 * (a long vector) waits dynamically for the exact rows it touches, which can lie a bounded distance AHEAD in the dispatch
 * order (enc_kernels.h rv_wait_rect).  All spins are bounded; an expired one poisons the row counter and sets errflag.
 */
/* wavefronts per SIMD the register allocation aims at.  Measured (gpurun_out/r3_wpe3): the two-wave pipeline at 3 (168 VGPRs, a few
 * spills, 6 rows per CU) beats 2 everywhere -- 1080p single slice 8.2 -> 8.9 M MB/s, 4K 10.8 -> 12.9 M; the one-wave kernel is
 * better off at 2 (256 VGPRs): at 3 it spills 131 registers. */
#ifndef H264E_WPE2
#define H264E_WPE2 3
#endif
#ifndef H264E_WPE1
#define H264E_WPE1 2
#endif
#ifndef H264E_WPEI
#define H264E_WPEI 4
#endif
/* OCC: wavefronts per SIMD the register allocation aims at.  The two-wave kernel exists twice: at 3 (164 VGPRs, nothing spilled) and at 4
 * (128 VGPRs, ~30 spilled); h264e_hip_submit picks per launch (more residency wins where a launch is bound by the rows in flight, fewer
 * spills where it is pure latency). */
template <int GEOM, int WAVES, int OCC>
__global__ void __launch_bounds__(64*WAVES, OCC) h264e_mb_kernel(h264e_geom_t G, const h264e_frame_task_t *tasks, const uint32_t *order)
{
    /* (the intra-only variant allocates the row state without the search-side buffers at its end: enc_mb.h ROWLDS_INTRA_BYTES) */
    __shared__ __attribute__((aligned(16))) unsigned char L_bytes[GEOM == GEOM_INTRA ? ROWLDS_INTRA_BYTES : sizeof(RowLds)];
    RowLds &L = *reinterpret_cast<RowLds *>(L_bytes);
    const int wv = WAVES == 2 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;     /* which wavefront of the workgroup */
#ifdef H264E_LDS_PAD
    /* diagnostic build (Makefile `halfres`): extra LDS per workgroup so that fewer workgroups fit a CU -- what single-stream throughput
     * does when the resident rows are halved with the macroblock latency unchanged (the price of a multi-wave workgroup) */
    __shared__ volatile char lds_pad[H264E_LDS_PAD];
    if (G.nmbx < 0) lds_pad[LANE] = 1;
#endif
    /* workgroups are dispatched in index order: `order` lists (job, row) by the step at which the row can start
     * (H264E_FRAME_LAG*job + 2*row), so the resident workgroups are the ones that can make progress */
    const uint32_t jr = order[blockIdx.x];
    const int job = (int)(jr >> 16), row = (int)(jr & 0xffffu);
    const h264e_frame_task_t &T = tasks[job];
    if (!T.active) return;
    const ChainG C = chain_view(*T.chain_desc);
    GLOBAL_AS int *errflag = (GLOBAL_AS int *)uniptr(T.errflag);

    if (row == G.nmby)
    {
        /* ---- finalizer (one wavefront; the second one of a two-wave workgroup has nothing to do here) */
        if (wv) return;
        int st = 0, seen = 0;
        GLOBAL_AS h264e_hostdone_t *hd = (GLOBAL_AS h264e_hostdone_t *)T.host_done;
        GLOBAL_AS h264e_walkrec_t *wout = (GLOBAL_AS h264e_walkrec_t *)T.walk_out;
        for (int r = G.nmby - 1; r >= 0 && !st; r--) st = poll_progress(C.progress + r, G.nmbx + 1, seen, G.spin_limit);
        if (st)
        {
            if (LANE == 0)
            {
                if (st == -1) *errflag = 1;
                if (T.walk_on_device && wout)
                {
                    /* the frames behind wait for this verdict: never leave them spinning */
                    wout->status = H264E_WALK_VOID; wout->first_bad = -1;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(&wout->flag, T.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (hd) { hd->walk_status = H264E_WALK_VOID; __hip_atomic_store(&hd->done, -T.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
            }
            return;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        int wstatus = 0, first_bad = -1;
        mv32 ws[2] = { T.exact_state[0], T.exact_state[1] };
        if (T.walk_on_device)
        {
            /* exact mv_clusters validation on the device: start from the verdict of the frame in front (same launch), walk this
             * frame's records, publish the verdict for the frame behind; a mismatch stops the whole launch through the abort word */
            if (T.walk_prev)
            {
                const GLOBAL_AS h264e_walkrec_t *wp = (const GLOBAL_AS h264e_walkrec_t *)T.walk_prev;
                int sf = 0;
                if (poll_progress(&wp->flag, T.launch_id, sf, G.spin_limit)) wstatus = H264E_WALK_VOID;
                else
                {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    if (uni(wp->status) != H264E_WALK_OK) wstatus = H264E_WALK_VOID;
                    else { ws[0] = (mv32)uni(wp->state_out[0]); ws[1] = (mv32)uni(wp->state_out[1]); }
                }
            }
            if (!wstatus)
            {
                first_bad = device_clusters_walk(G, T, C.mbrec + (size_t)T.frame_slot*G.nmb, ws, (GLOBAL_AS mv32 *)T.traj_out);
                wstatus = first_bad >= 0 ? H264E_WALK_BAD : H264E_WALK_OK;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (LANE == 0 && wout)
            {
                wout->state_out[0] = ws[0]; wout->state_out[1] = ws[1]; wout->status = wstatus; wout->first_bad = first_bad;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&wout->flag, T.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (wstatus != H264E_WALK_OK)
            {
                if (LANE == 0)
                {
                    if (wstatus == H264E_WALK_BAD && T.abort_word && !T.walk_quiet)
                        __hip_atomic_store((GLOBAL_AS int *)T.abort_word, T.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (hd)
                    {
                        hd->walk_status = wstatus; hd->first_bad = first_bad; hd->state_out[0] = ws[0]; hd->state_out[1] = ws[1];
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __hip_atomic_store(&hd->done, -T.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
                return;
            }
        }
        finalize_frame(G, C, T, (GLOBAL_AS int *)T.stepflags);
        if (hd)
        {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            uint32_t nal_bytes[H264E_MAX_SLICES], nal_total = 0;
            int exp_overflow = 0, in_device = 0;
            export_frame(G, C, T, nal_bytes, nal_total, exp_overflow, in_device);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (LANE == 0)
            {
                const GLOBAL_AS h264e_frameout_t &F = C.fout[T.frame_slot];
                hd->nbytes = nal_total; hd->all_skipped = F.all_skipped;
                hd->nslices = F.nslices; hd->in_device = in_device;
#pragma unroll
                for (int k = 0; k < H264E_MAX_SLICES; k++) hd->slice_nbytes[k] = nal_bytes[k];
                hd->clusters_moved = F.clusters_moved; hd->overflow = F.overflow | exp_overflow; hd->far_reads = F.far_reads;
                hd->walk_status = wstatus; hd->first_bad = first_bad; hd->state_out[0] = ws[0]; hd->state_out[1] = ws[1];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");                 /* system scope: the host reads these */
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&hd->done, T.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        return;
    }

    /* ---- macroblock row */
    if (row < T.first_row) return;          /* kept from the previous encode of this frame (its counter already says complete) */
    if (job == 0 && row == G.test_stall_row) return;    /* fault injection: a producer that never publishes (tests/test_gpu_failures.py) */
    if (wv == 0) row_begin(L, G, C, T, row);
    if (WAVES == 2) __syncthreads();         /* the one workgroup barrier: both wavefronts see the row's tables and initial state */
    const RowTask RT = rowtask_load(T);      /* the task's hot fields, once, in registers */
    int row0 = 0, row1 = G.nmby;            /* the slice (row band) this row belongs to */
    for (int k = 0; k < T.nslices; k++)
        if (row >= T.slice_row[k] && row < T.slice_row[k + 1]) { row0 = T.slice_row[k]; row1 = T.slice_row[k + 1]; }
    row0 = uni(row0); row1 = uni(row1);
    GLOBAL_AS int *my_progress = C.progress + row;

    if (WAVES == 1 || wv == 0)
    {
        /* ---- the wave that waits for the neighbours and searches (with one wave per row: does everything) */
        const GLOBAL_AS int *abort_word = (const GLOBAL_AS int *)uniptr(T.abort_word);
        const int launch_id = uni(T.launch_id);
        int seen = 0, seen_dep = 0;
        constexpr bool NARROW = GEOM == GEOM_NARROW;
        constexpr int DEP_ROWS = NARROW ? H264E_NARROW_DEP_ROWS : H264E_DEP_ROWS, DEP_COLS = NARROW ? H264E_NARROW_DEP_COLS : H264E_DEP_COLS;
        /* temporal wavefront: the reference window covers macroblock rows row-2 .. row+DEP_ROWS of the frame being referenced.
         * Inside one slice the lowest of them is the last to get there; with row-band slices the bands advance independently, so
         * every band the window touches is waited for at its lowest row inside the window (10 bits per row, lowest row first) */
        unsigned long long deps = 0;
        int ndeps = 0;
        if (T.dep_progress)
        {
            const int lo = imax(row - 2, 0), hi = imin(row + DEP_ROWS, G.nmby - 1);
            deps = (unsigned long long)hi; ndeps = 1;
            for (int k = T.nslices - 1; k >= 1; k--)
            {
                const int last = T.slice_row[k] - 1;            /* last row of slice k-1 */
                if (last >= lo && last < hi) { deps |= (unsigned long long)last << (10*ndeps); ndeps++; }
            }
            ndeps = uni(ndeps);
        }
        for (int x = 0; x < G.nmbx; x++)
        {
            /* consumer: relaxed sc1 polls, then sc1 loads of everything handed over */
            const int need = row > row0 ? imin(x + 2, G.nmbx) : 0;
            const int need_dep = RT.dep_progress ? imin(x + DEP_COLS, G.nmbx) : 0;     /* temporal wavefront: h264e_dev.h */
            int st = 0;
            /* the abort word lives in device memory (raised by a finalizer whose mv_clusters walk failed, or by the host): one L2 read per macroblock */
            if (abort_word && uni(__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == launch_id) st = -2;
            /* temporal dependency first, then the loads that only need it (input, reference window) ... */
            if (!st && seen_dep < need_dep)
            {
                int lowest = 0x7fffffff;
                for (int k = 0; k < ndeps && !st; k++)
                {
                    int sk = 0;
                    st = poll_progress((const GLOBAL_AS int *)RT.dep_progress + (int)((deps >> (10*k)) & 1023), need_dep, sk, G.spin_limit);
                    lowest = imin(lowest, sk);
                }
                seen_dep = lowest;
                if (!st) consumer_acquire();
            }
            /* two waves: macroblock x uses the hand-off buffer of x - 2, which the reconstruction wave must have left behind */
            STAMP(L, 13);
            if (WAVES == 2 && !st && x >= 2) st = lds_wait(&L.f_wdone, x - 1, &L.f_stop);
            STAMP(L, 23);
            if (!st) row_prefetch<GEOM>(L, G, RT, row, x);
            STAMP(L, 0);
            /* ... so that their latency overlaps with the wait for the row above */
            if (!st && seen < need)
            {
                st = poll_progress(C.progress + (row - 1), need, seen, G.spin_limit);
                if (!st) consumer_acquire();
            }
            STAMP(L, 13);
            if (!st) load_top(L, L.mb[x & 1], G, C.bottom + (size_t)(row - 1)*G.nmbx, C.pend + (size_t)(row - 1)*G.nmbx, x, row > row0);
            /* two waves: the search of x starts from the predictor context the decision of x - 1 leaves behind */
            STAMP(L, 0);
            if (WAVES == 2 && !st && x >= 1) st = lds_wait(&L.f_decided, x, &L.f_stop);
            STAMP(L, 6);
            if (st)
            {
                /* stop: leave poison in this row's counter so everything behind it stops too (-1 failure, -2 abort); with two waves the
                 * reconstruction wave is the one that publishes, so it also leaves the poison (after what it is publishing right now) */
                if (WAVES == 2) { flag_set(&L.f_stop, st); return; }
                if (LANE == 0)
                {
                    if (st == -1) *errflag = 1;
                    __hip_atomic_store(my_progress, st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return;
            }
            if (WAVES == 1) row_step<GEOM>(L, G, C, RT, row, x, row0, row1);
            else mb_search<GEOM>(L, L.mb[x & 1], G, RT, row, x, row0, SearchSignals{ &L, x + 1 });
            {
                /* a far reference read of this macroblock waited in vain (rv_wait_rect): what it encoded is not trustworthy */
                const int ff = uni(L.far_fail[0]) | (WAVES == 1 ? uni(L.far_fail[1]) : 0);
                if (ff)
                {
                    if (WAVES == 2) { flag_set(&L.f_stop, ff < -2 ? -2 : ff); return; }
                    if (LANE == 0)
                    {
                        if (ff == -1) *errflag = 1;
                        __hip_atomic_store(my_progress, ff < -2 ? -2 : ff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    return;
                }
            }
            if (WAVES == 2) { flag_set(&L.f_inter, x + 1); STAMP(L, 14); continue; }
            /* producer: every handed-off byte was stored write-through (sc1, wave.h cstore*): drain them, then the counter --
             * no agent-scope release (L2 write-back) needed */
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (LANE == 0) __hip_atomic_store(my_progress, x + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            STAMP(L, 14);
        }
        if (WAVES == 2)
        {
#ifdef H264E_STAMPS
            if (LANE < 32 && C.prof) atomicAdd(C.prof + LANE, L.prof[0][LANE]);
#endif
            return;
        }
    } else
    {
        /* ---- the reconstruction wave of the two-wave pipeline: intra candidates + decision of x, then transform / CAVLC / deblocking /
         * stores of x while the search wave is already on x + 1 */
        for (int x = 0; x < G.nmbx; x++)
        {
            MbBuf &B = L.mb[x & 1];
            MbCtx m;
            mb_ctx_init<GEOM>(m, L, G, RT, row, x, row0, 1);
            int ff = 0;
            if (!mb_intra_decide(L, B, m, RT, InterFromSearchWave{ &L, x + 1 })) ff = uni(flag_get(&L.f_stop));
            else
            {
                flag_set(&L.f_decided, x + 1);
                mb_recon_write<GEOM>(L, B, m, G, C, RT, row, x, row0, row1);
                ff = uni(L.far_fail[1]);
                if (ff) { ff = ff < -2 ? -2 : ff; flag_set(&L.f_stop, ff); }
            }
            if (ff)
            {
                if (LANE == 0)
                {
                    if (ff == -1) *errflag = 1;
                    __hip_atomic_store(my_progress, ff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (LANE == 0) __hip_atomic_store(my_progress, x + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            flag_set(&L.f_wdone, x + 1);
            STAMP(L, 14);
        }
    }
    row_end(L, G, C, row);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (LANE == 0) __hip_atomic_store(my_progress, G.nmbx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   /* row buffer + meta complete */
}

#endif

#ifndef H264E_EMU
__global__ void __launch_bounds__(64) h264e_nal_escape_selftest_kernel(uint8_t *dst, uint32_t cap, const uint8_t *src, uint32_t n, uint32_t *out)
{
    int overflow = 0;
    const uint32_t w = nal_escape_copy((GLOBAL_AS uint8_t *)dst, cap, (const GLOBAL_AS uint8_t *)src, n, overflow);
    if (threadIdx.x == 0) { out[0] = w; out[1] = (uint32_t)overflow; }
}
#endif

/* ------------------------------------------------------------------ per-stage test hook (tests/test_stages.py)
 * Runs ONE of the macroblock pipeline's wave-level stages on caller-supplied operands, so that each can be compared with the
 * reference's own function of the same stage (tests/golden/stages.json, made by oracle/stage_harness.c):
 *   1 SAD quadrants (wave_sad_ref_q)           in: picture 64x64 | block 16x16         args: x, y, window      out: int32 sad4[4], sum
 *   2 luma quarter-sample (wave_interp_luma)   in: picture 64x64                       args: x, y, w, h, dx, dy, window   out: 16x16 (stride 16)
 *   3 chroma bilinear (wave_interp_chroma)     in: picture 64x64 (used as U and V)     args: x, y, w, h, dx, dy           out: 16x16: U cols 0-7, V cols 8-15
 *   4 transform/quant/dequant/recon            in: inp 256 | pred 256 | qdat 42 x u16  args: mode              out: int32 nz, dcflag | qblk_t q[16] | i16 dc[16] | i16 lev[16] | recon 256
 *   5 CAVLC block (cavlc_block)                in: int16 coef[16]                      args: first, maxn, nctx out: int32 nnz, nbits | bytes
 *   8 motion search of one partition (diamond)  in: picture 96x96 | macroblock 16x16      args: px, py, w, h, mv x/y, pred x/y, min_sad, qp, speed, range[4], limit[4], window
 *                                                out: int32 cost, mv x, mv y | prediction 16x16 (stride 16)
 *   7 deblock one macroblock (wave_deblock)     in: luma tile 20x24 | U tile 10x12 | V tile 10x12 | bs 32   args: qp, qp_left, qp_top   out: the three tiles
 *   6 intra 4x4 mode choice (wave_i4_choose)   in: edge 13 (L3..L0, UL, U0..U7) | block 4x4 stride 16   args: avail, mpred, penalty   out: int32 mode, cost | prediction 4x4 stride 16
 * `window` = 1 reads the reference samples through the LDS window like the macroblock loop, 0 through the HBM path.
 */
struct StageLds
{
    alignas(16) uint8_t win[WIN_W*WIN_STRIDE + 16];
    alignas(16) uint8_t a[256], b[256], o[256];
    alignas(16) qblk_t q[16];
    alignas(4) int16_t dc[16], lev[16], coef[16];
    alignas(4) uint16_t qdat[42];
    CavlcTab ct;
    I4Scratch i4s;
    DfTab df;
    alignas(16) uint8_t yt[20*YT_STRIDE];
    alignas(16) uint8_t ctile[2][10*CT_STRIDE];
    alignas(4) uint8_t bs[32];
    alignas(4) uint8_t nb[5*24];         /* the block's neighbourhood as intra4_choose keeps it: row stride 24, block at row 1, column 4 */
};
#define STAGE_IN_MAX (96*96 + 256)
#define STAGE_NARGS 24
#define STAGE_OUT_MAX (16 + 16*64 + 32 + 32 + 256)
DEV void stage_selftest(StageLds &S, RowLds &L, int stage, const GLOBAL_AS uint8_t *in, const int *args, GLOBAL_AS uint8_t *out)
{
    int a[STAGE_NARGS];
    for (int i = 0; i < STAGE_NARGS; i++) a[i] = uni(args[i]);
    Plane P = { (const gu8 *)in, 64, 64, 64 };
    RefView R;
    R.P = P; R.win = (const lu8 *)S.win; R.has_win = 0; R.wx0 = 0; R.wy0 = 0; R.dep = 0; R.nmbx = 4; R.nmby = 4; R.vw = WIN_W; R.vh = WIN_W;
    R.far = 0; R.fail = 0; R.slice_row = 0; R.nslices = 0; R.spin_limit = 0;
    GLOBAL_AS int32_t *oi = (GLOBAL_AS int32_t *)out;
    if (stage == 1 || stage == 2)
    {
        const int x = a[0], y = a[1], window = stage == 1 ? a[2] : a[6];
        if (window) { R.has_win = 1; R.wx0 = x - WIN_M; R.wy0 = y - WIN_M; wave_load_window(S.win, P, R.wx0, R.wy0, 0); wave_sync(); }
        if (stage == 1)
        {
            int s4[4];
            WAVE_FOR(l) { lds32_store(S.b + 4*l, gload32((const gu8 *)in + 4096 + 4*l)); }
            wave_sync();
            const int tot = wave_sad_ref_q(R, x, y, S.b, s4);
            if (wave_lane() == 0) { oi[0] = s4[0]; oi[1] = s4[1]; oi[2] = s4[2]; oi[3] = s4[3]; oi[4] = tot; }
        } else
        {
            WAVE_FOR(l) { lds32_store(S.o + 4*l, 0u); }
            wave_sync();
            wave_interp_luma(R, 0, 0, mvmk(4*x + a[4], 4*y + a[5]), a[2], a[3], S.o);
            wave_sync();
            WAVE_FOR(l) { gstore32((gu8 *)out + 4*l, lds32(S.o + 4*l)); }
        }
    } else if (stage == 3)
    {
        WAVE_FOR(l) { lds32_store(S.o + 4*l, 0u); }
        wave_sync();
        wave_interp_chroma(R, P, P, 0, 0, mvmk(8*a[0] + a[4], 8*a[1] + a[5]), a[2], a[3], S.o);
        wave_sync();
        WAVE_FOR(l) { gstore32((gu8 *)out + 4*l, lds32(S.o + 4*l)); }
    } else if (stage == 4)
    {
        const int mode = a[0], side = mode >> 1;
        WAVE_FOR(l)
        {
            lds32_store(S.a + 4*l, gload32((const gu8 *)in + 4*l));
            lds32_store(S.b + 4*l, gload32((const gu8 *)in + 256 + 4*l));
            if (l < 21) lds32_store((uint8_t *)S.qdat + 4*l, gload32((const gu8 *)in + 512 + 4*l));
            if (l < 8) { lds32_store((uint8_t *)S.dc + 4*l, 0u); lds32_store((uint8_t *)S.lev + 4*l, 0u); }
            for (int k = l; k < 256; k += 64) lds32_store((uint8_t *)S.q + 4*k, 0u);
        }
        wave_sync();
        unsigned nz = wave_xform_quant(S.a, S.b, mode, S.q, S.dc, S.qdat);
        int dcflag = 0;
        wave_sync();
        if (mode == QMODE_I16) quant_luma_dc(S.q, S.dc, S.lev, S.qdat);
        if (mode == QMODE_CHROMA) dcflag = quant_chroma_dc(S.q, S.dc, S.lev, S.qdat);
        wave_sync();
        /* the operands as the reference has them in front of the reconstruction */
        WAVE_FOR(l)
        {
            for (int k = l; k < 256; k += 64) gstore32((gu8 *)out + 8 + 4*k, lds32((const uint8_t *)S.q + 4*k));
            if (l < 8) { gstore32((gu8 *)out + 8 + 1024 + 4*l, lds32((const uint8_t *)S.dc + 4*l)); gstore32((gu8 *)out + 8 + 1056 + 4*l, lds32((const uint8_t *)S.lev + 4*l)); }
        }
        /* reconstruction as mb_write / intra4_choose call it (h264-lab.h:4428-4433, 4468-4488, 4809-4811) */
        WAVE_FOR(l) { lds32_store(S.o + 4*l, lds32(S.b + 4*l)); }
        wave_sync();
        if (mode == QMODE_INTER) wave_recon(S.o, 16, S.b, S.q, 4, nz << 16);
        else if (mode == QMODE_I16) wave_recon(S.o, 16, S.b, S.q, 4, 0xffffu << 16);
        else if (mode == QMODE_I4) { if (nz & 1) wave_recon(S.o, 16, S.b, S.q, 1, 0x80000000u); }
        else if (dcflag | (int)nz)
        {
            unsigned m = nz;
            if (dcflag)
            {
                WAVE_FOR(l) { if (l < 60) { const int b4 = l/15, i = 1 + l % 15; if (~nz & (8u >> b4)) S.q[b4].dq[i] = 0; } }
                wave_sync();
                m = 15;
            }
            wave_recon(S.o, 16, S.b, S.q, 2, m << 28);
        }
        wave_sync();
        WAVE_FOR(l) { gstore32((gu8 *)out + 8 + 1088 + 4*l, lds32(S.o + 4*l)); }
        if (wave_lane() == 0) { oi[0] = (int32_t)nz; oi[1] = dcflag; }
        (void)side;
    } else if (stage == 5)
    {
        cavlc_tab_load(S.ct);
        WAVE_FOR(l) { if (l < 8) lds32_store((uint8_t *)S.coef + 4*l, gload32((const gu8 *)in + 4*l)); }
        wave_sync();
        BitW b;
        b.acc = 0; b.nacc = 0; b.pos = 0; b.cap = 60; b.overflow = 0; b.buf = (GLOBAL_AS uint32_t *)(out + 8);
        const int nnz = cavlc_block(b, S.ct, S.coef, a[0], a[1], a[2]);
        const uint32_t nbits = bw_bits(b);
        if (b.nacc) bw_put(b, 32 - b.nacc, 0);
        if (wave_lane() == 0) { oi[0] = nnz; oi[1] = (int32_t)nbits; }
    } else if (stage == 6)
    {
        WAVE_FOR(l)
        {
            for (int k = l; k < 144; k += 64) S.i4s.lut[k] = k_i4_lut[k/16][k%16];
            if (l < 13)
            {
                const uint8_t e = in[l];
                if (l < 4) S.nb[24*(4 - l) + 3] = e;            /* L3..L0: the column left of the block, bottom-up */
                else S.nb[3 + (l - 4)] = e;                     /* UL, U0..U7: the row above */
            }
            if (l < 16) lds32_store(S.a + 4*l, gload32((const gu8 *)in + 16 + 4*l));
            if (l < 16) lds32_store(S.o + 4*l, 0u);
        }
        wave_sync();
        const int res = wave_i4_choose(S.a, S.o, a[0], S.nb + 4, S.nb + 24 + 3, 24, a[1], a[2], S.i4s);
        wave_sync();
        WAVE_FOR(l) { if (l < 16) gstore32((gu8 *)out + 8 + 4*l, lds32(S.o + 4*l)); }
        if (wave_lane() == 0) { oi[0] = res & 15; oi[1] = res >> 4; }
    } else if (stage == 7)
    {
        const int ny = 20*YT_STRIDE, nc = 10*CT_STRIDE;
        df_tab_load(S.df);
        WAVE_FOR(l)
        {
            for (int k = l; k < ny; k += 64) S.yt[k] = in[k];
            for (int k = l; k < nc; k += 64) { S.ctile[0][k] = in[ny + k]; S.ctile[1][k] = in[ny + nc + k]; }
            if (l < 32) S.bs[l] = in[ny + 2*nc + l];
        }
        wave_sync();
        wave_deblock(S.yt, S.ctile[0], S.ctile[1], S.bs, a[0], a[1], a[2], S.df);
        wave_sync();
        WAVE_FOR(l)
        {
            for (int k = l; k < ny; k += 64) out[k] = S.yt[k];
            for (int k = l; k < nc; k += 64) { out[ny + k] = S.ctile[0][k]; out[ny + nc + k] = S.ctile[1][k]; }
        }
    } else if (stage == 8)
    {
        /* the macroblock at (32,32) of a 96x96 reference picture, as row_step sets a macroblock up for inter_choose */
        h264e_geom_t Gs;
        MbCtx m;
        Plane P8 = { (const gu8 *)in, 96, 96, 96 };
        Gs.width = Gs.W = 96; Gs.height = Gs.H = 96; Gs.nmbx = Gs.nmby = 6; Gs.nmb = 36; Gs.cropping = 0;
        Gs.lim_x0 = a[15]; Gs.lim_y0 = a[16]; Gs.lim_x1 = a[17]; Gs.lim_y1 = a[18];
        m.G = &Gs; m.speed = a[10]; m.slice_type = 0; m.x = 2; m.y = 2; m.num = 14; m.qp = a[9];
        m.lambda_mv = k_lambda_mv_q4[a[9]];
        m.rv = R; m.rv.P = P8; m.rv.nmbx = 6; m.rv.nmby = 6;
        if (a[19]) { m.rv.has_win = 1; m.rv.win = (const lu8 *)L.win; m.rv.wx0 = 32 - WIN_M; m.rv.wy0 = 32 - WIN_M; wave_load_window(L.win, P8, m.rv.wx0, m.rv.wy0, 0); }
        WAVE_FOR(l) { lds32_store(L.mb[0].inp + 4*l, gload32((const gu8 *)in + 96*96 + 4*l)); lds32_store(L.gtest[0] + 4*l, 0u); }
        wave_sync();
        const rect_t range = { a[11], a[12], a[13], a[14] };
        /* the search is lane-group code (wave.h): group a[20] runs it, the other three idle */
        GRP_EACH(grp)
        {
            if (grp == (a[20] & 3))
            {
                mv32 mv = mvmk(a[4], a[5]);
                const int cost = diamond_g(L, L.mb[0], m, a[0], a[1], mv, range, mvmk(a[6], a[7]), a[8], a[2], a[3], L.gtest[0] + 16*a[1] + a[0], L.gscr);
                L.gcost[0] = cost; L.gcost[1] = mvx(mv); L.gcost[2] = mvy(mv);
            }
        }
        wave_sync();
        WAVE_FOR(l) { gstore32((gu8 *)out + 16 + 4*l, lds32(L.gtest[0] + 4*l)); }
        if (wave_lane() == 0) { oi[0] = L.gcost[0]; oi[1] = L.gcost[1]; oi[2] = L.gcost[2]; }
    }
}
#ifndef H264E_EMU
__global__ void __launch_bounds__(64) h264e_stage_selftest_kernel(int stage, const uint8_t *in, const int *args, uint8_t *out)
{
    __shared__ StageLds S;
    __shared__ RowLds L;
    stage_selftest(S, L, stage, (const GLOBAL_AS uint8_t *)in, args, (GLOBAL_AS uint8_t *)out);
}
#endif

/* synth_v1 generator, one sample per call */
DEV uint32_t sv_h32(uint32_t a)
{
    a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
    return a;
}
DEV int sv_lattice(int32_t ix, int32_t iy, uint32_t seed) { return (int)(sv_h32((uint32_t)ix*0x9E3779B1u ^ (uint32_t)iy*0x85EBCA77u ^ seed) & 255); }
DEV int sv_tex(int32_t X, int32_t Y, uint32_t seed, int lg)
{
    int32_t c = 1 << lg, ix = X >> lg, iy = Y >> lg, fx = X & (c - 1), fy = Y & (c - 1);
    int32_t a = sv_lattice(ix, iy, seed), b = sv_lattice(ix + 1, iy, seed), cc = sv_lattice(ix, iy + 1, seed), d = sv_lattice(ix + 1, iy + 1, seed);
    int32_t top = a*(c - fx) + b*fx, bot = cc*(c - fx) + d*fx;
    return (top*(c - fy) + bot*fy + (1 << (2*lg - 1))) >> (2*lg);
}
DEV uint8_t sv_sample(int w, int h, int t, uint32_t seed, int idx)
{
    const int32_t OFF = 1 << 20;
    if (idx < w*h)
    {
        int x = idx % w, y = idx / w, fw = w/8 > 32 ? w/8 : 32, fh = h/6 > 32 ? h/6 : 32;
        int fx0 = (w/2 + ((10*t) >> 2)) % (w - fw), fy0 = h/3, v;
        if (x >= fx0 && x < fx0 + fw && y >= fy0 && y < fy0 + fh) v = sv_tex(4*x - 10*t + OFF, 4*y + OFF, seed + 1, 5);
        else v = (sv_tex(4*x + 5*t + OFF, 4*y + 3*t + OFF, seed, 6)*3 >> 2) + 32;
        v += (int)(sv_h32((uint32_t)x ^ ((uint32_t)y << 12) ^ ((uint32_t)t << 24) ^ (uint32_t)(seed*7919u)) % 5) - 2;
        return (uint8_t)clip255(v);
    }
    idx -= w*h;
    const int cw = w/2, ch = h/2, pl = idx >= cw*ch;
    if (pl) idx -= cw*ch;
    int x = idx % cw, y = idx / cw;
    if (!pl) return (uint8_t)(128 + ((sv_tex(8*x + 5*t + OFF, 8*y + 3*t + OFF, seed + 2, 7) - 128) >> 2));
    return (uint8_t)(128 - ((sv_tex(8*x + 5*t + OFF, 8*y + 3*t + OFF, seed + 3, 7) - 128) >> 3));
}

#ifndef H264E_EMU
__global__ void h264e_synth_kernel(uint8_t *dst, int w, int h, int t, uint32_t seed)
{
    const int n = w*h*3/2;
    for (int i = (int)(blockIdx.x*blockDim.x + threadIdx.x); i < n; i += (int)(gridDim.x*blockDim.x))
        dst[i] = sv_sample(w, h, t, seed, i);
}
#endif

/* ------------------------------------------------------------------ host side: pool */

#ifdef H264E_EMU
#define DEVCALL(x) (x)
static int dev_malloc(void **p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? 0 : -1; }
static void dev_free(void *p) { free(p); }
#define H2D(d, s, n) memcpy(d, s, n)
#define D2H(d, s, n) memcpy(d, s, n)
#else
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) FAIL("%s: %s", #x, hipGetErrorString(e_)); } while (0)
static int dev_malloc(void **p, size_t n) { return hipMalloc(p, n ? n : 1) == hipSuccess ? 0 : -1; }
static void dev_free(void *p) { if (p) (void)hipFree(p); }
#endif

#ifdef H264E_EMU
static int host_malloc(void **p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? 0 : -1; }
static void host_free(void *p) { free(p); }
#else
/* pinned, device-mapped, coherent host memory: the kernel writes results here while it runs, the host polls it */
static int host_malloc(void **p, size_t n)
{
    if (hipHostMalloc(p, n ? n : 1, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return -1;
    memset(*p, 0, n ? n : 1);
    return 0;
}
static void host_free(void *p) { if (p) (void)hipHostFree(p); }
#endif

#define TASK_RING 128
static int imin_h(int a, int b) { return a < b ? a : b; }

/* One launch at a time per device, process-wide.  The macroblock kernel's forward-progress argument (every workgroup waits for
 * workgroups dispatched before it, which are resident or finished) assumes the launch has the device's wave slots to itself: two
 * such launches side by side can fill the slots with waiting workgroups of one while the workgroups they wait for sit undispatched
 * behind the other's (measured: "bounded spin expired" with 3-4 concurrent clip encoders, tools/multi_clip_probe.py).  A pool takes
 * its device's lock with its first submit and gives it back when its launches have drained (h264e_hip_sync / release / destroy). */
#ifndef H264E_EMU
#include <pthread.h>
#define H264E_MAX_DEVICES 64
static pthread_mutex_t g_device_lock[H264E_MAX_DEVICES] = { PTHREAD_MUTEX_INITIALIZER };
static pthread_once_t g_device_lock_once = PTHREAD_ONCE_INIT;
static void device_locks_init(void) { for (int i = 0; i < H264E_MAX_DEVICES; i++) pthread_mutex_init(&g_device_lock[i], 0); }
#endif

/*
 * One encoder PROCESS per device.  The launch lock above only orders the launches of one process; a second process on the same GPU
 * would put its persistent launches next to ours (bounded spins expire, launches are repeated: slow, never wrong).  The first pool a
 * process creates on a device therefore takes an advisory lock on a file named after the device's PCI bus id and keeps it until its
 * last pool on that device is gone; a second process fails fast with a message that says who holds the device.
 * H264E_SHARE_DEVICE=1 skips the guard (e.g. to run two small encoders side by side on purpose).
 */
#ifndef H264E_EMU
#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>
static pthread_mutex_t g_guard_mu = PTHREAD_MUTEX_INITIALIZER;
static int g_guard_fd[H264E_MAX_DEVICES], g_guard_pools[H264E_MAX_DEVICES], g_guard_init;
static int process_guard_acquire(int device)
{
    const char *share = getenv("H264E_SHARE_DEVICE");
    int rc = 0;
    if ((share && atoi(share) == 1) || device < 0 || device >= H264E_MAX_DEVICES) return 0;
    pthread_mutex_lock(&g_guard_mu);
    if (!g_guard_init) { for (int i = 0; i < H264E_MAX_DEVICES; i++) g_guard_fd[i] = -1; g_guard_init = 1; }
    if (g_guard_pools[device]++ == 0)
    {
        char bus[64] = "", path[160];
        if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess || !bus[0]) snprintf(bus, sizeof(bus), "dev%d", device);
        for (char *q = bus; *q; q++) if (*q == ':' || *q == '.' || *q == '/') *q = '_';
        snprintf(path, sizeof(path), "%s/h264e_mi355x_%s.lock", getenv("H264E_LOCK_DIR") ? getenv("H264E_LOCK_DIR") : "/tmp", bus);
        const int fd = open(path, O_RDWR | O_CREAT, 0666);
        if (fd >= 0)
        {
            if (flock(fd, LOCK_EX | LOCK_NB))
            {
                char who[32] = "";
                const ssize_t n = read(fd, who, sizeof(who) - 1);
                if (n > 0) { who[n] = 0; for (char *q = who; *q; q++) if (*q == '\n') *q = 0; }
                snprintf(g_err, sizeof(g_err), "device %d (%s) is in use by another encoder process (pid %s): the macroblock kernel needs the device's wave slots to itself -- "
                         "give each process its own GPU, or encode several streams in ONE process (H264E_clip_encode_multi); H264E_SHARE_DEVICE=1 overrides", device, bus, who[0] ? who : "?");
                close(fd);
                g_guard_pools[device]--;
                rc = -1;
            } else
            {
                char me[32];
                const int n = snprintf(me, sizeof(me), "%ld\n", (long)getpid());
                if (ftruncate(fd, 0) == 0 && write(fd, me, (size_t)n) != n) { /* the pid is informational */ }
                g_guard_fd[device] = fd;
            }
        }       /* (no lock directory: no guard) */
    }
    pthread_mutex_unlock(&g_guard_mu);
    return rc;
}
static void process_guard_release(int device)
{
    if (device < 0 || device >= H264E_MAX_DEVICES) return;
    pthread_mutex_lock(&g_guard_mu);
    if (g_guard_init && g_guard_pools[device] > 0 && --g_guard_pools[device] == 0 && g_guard_fd[device] >= 0) { close(g_guard_fd[device]); g_guard_fd[device] = -1; }
    pthread_mutex_unlock(&g_guard_mu);
}
#endif

struct h264e_hip_group;
typedef struct h264e_hip_group h264e_hip_group_t;

struct h264e_hip_pool
{
    int device, nchains, frames_resident, slots;
    h264e_geom_t G;
    size_t frame_bytes;
    uint8_t *clip;                       /* device: resident input frames, packed I420 */
    h264e_chain_dev_t *chains_host;      /* host mirror of the device descriptors */
    h264e_chain_dev_t *chains_dev;
    h264e_frame_task_t *tasks_dev;       /* ring of TASK_RING task arrays */
    int *progress_all;
    int *errflag;
    uint32_t *order;                     /* device [nchains*(nmby+1)] (job << 16) | row in dispatch order of the current launch shape */
    uint32_t *order_host;                /* host copy being built (build_order) */
    int order_jobs, order_narrow;        /* the launch shape `order` holds: jobs, window geometry (-1: none yet) */
    int *stepflags;                      /* [nchains][2]: {clusters_moved, overflow} of the last step, one read per step */
    /* streaming: per chain slot, host-mapped result buffers the finalizer workgroups fill while the launch runs */
    h264e_hostdone_t *host_done;         /* [nchains] */
    uint8_t **host_rbsp;                 /* [nchains], each host_rbsp_cap bytes */
    h264e_hip_mbrec_t **host_mbrec;      /* [nchains], each nmb records */
    uint32_t host_rbsp_cap;
    int *abort_word;                     /* host-mapped: source of the host's own abort request */
    int *abort_dev;                      /* device: the word the kernel polls */
    h264e_walkrec_t *walkrec;            /* device [nchains] */
    int32_t **traj_dev;                  /* per chain: two [nmb][2] trajectory buffers behind each other */
    int *traj_cur;                       /* per chain: which of the two holds the latest device walk */
    unsigned long long *ssd_dev;         /* [nchains][3] sums of squared differences (h264e_hip_ssd_frames) */
    uint8_t *heap; size_t heap_bytes;    /* ONE device allocation; every device buffer of the pool is carved out of it */
    uint8_t *hheap; size_t hheap_bytes;  /* ONE host-mapped allocation for the streaming mirrors */
    int launch_counter;
    int holds_device;                    /* this pool has launches in flight and owns its device's launch lock */
    int *slot_launch;                    /* per chain slot: launch id of its current job */
    int32_t **clu_dev;                   /* per chain: optional per-macroblock mv_clusters array */
    int *ref_sel;                        /* per chain */
    int ring_pos, pending;
    int profile, prof_launches;
    int guarded;                         /* this pool counts in its device's process guard */
    struct h264e_hip_group *group;       /* launch group this pool's submits go through, or NULL */
    int group_round;                     /* the group round of its last submit */
    int waves;                           /* wavefronts per macroblock row forced by H264E_WAVES (1 or 2); 0 = chosen per launch (h264e_hip_submit) */
    int test_upload_fail_at, async_uploads;     /* fault injection (H264E_TEST_KNOBS): the n-th asynchronous upload of this pool fails */
    double prof_mb_ms, prof_splice_ms;
#ifndef H264E_EMU
    hipStream_t stream;
    hipStream_t copy_stream;             /* uploads that overlap with kernels on `stream` */
    hipStream_t abort_stream;            /* carries nothing but abort requests (h264e_hip_stream_abort) */
    hipEvent_t ev_t0, ev_t1, ev_prep;
    hipEvent_t ev[TASK_RING][3];         /* per pending submit: before / between / after the two kernels */
    int ev_pending;
#endif
};

/* launch groups (see h264e_hip_group_create below) */
#define H264E_GROUP_MAX 8
struct h264e_hip_group
{
    int device, nmembers, arrived, round, failed;
    h264e_hip_pool_t *member[H264E_GROUP_MAX];
    /* what each member wants launched this round */
    h264e_frame_task_t *pend_tasks[H264E_GROUP_MAX];
    int pend_jobs[H264E_GROUP_MAX], pend_narrow[H264E_GROUP_MAX], pend_waves[H264E_GROUP_MAX], pend[H264E_GROUP_MAX];
#ifndef H264E_EMU
    pthread_mutex_t mu;
    pthread_cond_t cv;
    hipStream_t stream;
    hipEvent_t ev_done, ev_t0, ev_t1;
    h264e_frame_task_t *tasks_dev; size_t tasks_cap;
    uint32_t *order_dev; size_t order_cap;
#endif
};

extern "C" int h264e_hip_device_count(void)
{
#ifdef H264E_EMU
    return 1;
#else
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
#endif
}

static void device_acquire(h264e_hip_pool_t *p)
{
#ifndef H264E_EMU
    if (p->holds_device) return;
    pthread_once(&g_device_lock_once, device_locks_init);
    pthread_mutex_lock(&g_device_lock[(unsigned)p->device % H264E_MAX_DEVICES]);
#endif
    p->holds_device = 1;
}
static void device_release(h264e_hip_pool_t *p)
{
    if (!p->holds_device) return;
    p->holds_device = 0;
#ifndef H264E_EMU
    pthread_mutex_unlock(&g_device_lock[(unsigned)p->device % H264E_MAX_DEVICES]);
#endif
}

extern "C" void h264e_hip_pool_destroy(h264e_hip_pool_t *p)
{
    if (!p) return;
#ifndef H264E_EMU
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    if (p->copy_stream) (void)hipStreamSynchronize(p->copy_stream);
    if (p->abort_stream) (void)hipStreamSynchronize(p->abort_stream);
#endif
    device_release(p);
    host_free(p->hheap);
    free(p->host_rbsp); free(p->host_mbrec); free(p->slot_launch); free(p->order_host);
    dev_free(p->heap);
#ifndef H264E_EMU
    if (p->stream)
    {
        for (int i = 0; i < TASK_RING; i++) for (int k = 0; k < 3; k++) (void)hipEventDestroy(p->ev[i][k]);
        (void)hipEventDestroy(p->ev_t0); (void)hipEventDestroy(p->ev_t1); (void)hipEventDestroy(p->ev_prep);
        (void)hipStreamDestroy(p->stream);
        if (p->copy_stream) (void)hipStreamDestroy(p->copy_stream);
        if (p->abort_stream) (void)hipStreamDestroy(p->abort_stream);
    }
#endif
    free(p->chains_host); free(p->clu_dev); free(p->ref_sel); free(p->traj_dev); free(p->traj_cur);
#ifndef H264E_EMU
    if (p->guarded) process_guard_release(p->device);
#endif
    free(p);
}

#ifndef H264E_EMU
/* Dispatch order of a launch of `jobs` jobs: (job, row) sorted by the step at which the row can start when consecutive jobs are
 * consecutive frames of one stream (lag*job + 2*row: a counting sort); every workgroup still only waits for workgroups that precede it
 * in this order (far reads: a bounded distance ahead).  Built for the number of jobs a launch really has, so that a pool with many
 * slots does not dispatch thousands of empty workgroups with every short launch.
 * H264E_XCD_BANDS=N (experiment, profiles/r02_xcd_bands.txt): workgroups are dealt round-robin over the 8 XCDs (MI355X_MICROARCH.md:
 * blocks b and b+8 share one), so the order is additionally arranged so that a macroblock row lands on the XCD of its band of rows
 * (row*N/nmby mod 8): the rows whose reference windows overlap then share an L2. */
static int build_order(h264e_hip_pool_t *p, int jobs, int narrow)
{
    const h264e_geom_t &G = p->G;
    const int rows = G.nmby + 1, total = jobs*rows, lag = narrow ? H264E_NARROW_FRAME_LAG : H264E_FRAME_LAG, maxkey = lag*(jobs - 1) + 2*(rows - 1);
    /* measured with the two-wave kernel (gpurun_out/r3_bands1): 8 bands halve FETCH_SIZE everywhere (1080p: 1896 -> 919 MB per launch; fetch +
     * write 2827 -> 1744 MB) -- and cost 2-4 % speed at 1080p and below, but GAIN 8 % at 4K, where a frame's rows no longer fit the L2s at
     * random: on by default from 4K up */
    const int bands = getenv("H264E_XCD_BANDS") ? atoi(getenv("H264E_XCD_BANDS")) : (G.nmb >= 30000 ? 8 : 0);
    uint32_t *ord = p->order_host;
    int *start = (int *)calloc((size_t)maxkey + 2, sizeof(int));
    uint32_t *tmp = bands ? (uint32_t *)malloc(sizeof(uint32_t)*(size_t)total) : ord;
    if (!start || !tmp) { free(start); if (bands) free(tmp); return -1; }
    for (int job = 0; job < jobs; job++) for (int r = 0; r < rows; r++) start[lag*job + 2*r + 1]++;
    for (int k = 0; k <= maxkey; k++) start[k + 1] += start[k];
    for (int job = 0; job < jobs; job++) for (int r = 0; r < rows; r++) tmp[start[lag*job + 2*r]++] = ((uint32_t)job << 16) | (uint32_t)r;     /* ties: by job */
    free(start);
    if (bands)
    {
        /* eight queues in key order, one per XCD; slot i takes the head of queue i % 8 (or, when that one has run dry, the head with
         * the smallest key) */
        int head[8], cnt[8] = { 0 }, n = 0;
        uint32_t *q = (uint32_t *)malloc(sizeof(uint32_t)*8*(size_t)total);
        if (!q) { free(tmp); return -1; }
        for (int i = 0; i < total; i++)
        {
            const int row = (int)(tmp[i] & 0xffffu), x = (row >= G.nmby ? bands - 1 : imin_h(bands - 1, row*bands/G.nmby)) & 7;      /* band b -> XCD b % 8 */
            q[(size_t)x*total + cnt[x]++] = tmp[i];
        }
        for (int x = 0; x < 8; x++) head[x] = 0;
        for (int i = 0; i < total; i++)
        {
            int x = i & 7;
            if (head[x] >= cnt[x])
            {
                long best = -1; x = -1;
                for (int y = 0; y < 8; y++)
                    if (head[y] < cnt[y])
                    {
                        const uint32_t jr2 = q[(size_t)y*total + head[y]];
                        const long key = (long)lag*(long)(jr2 >> 16) + 2*(long)(jr2 & 0xffffu);
                        if (x < 0 || key < best) { best = key; x = y; }
                    }
            }
            ord[n++] = q[(size_t)x*total + head[x]++];
        }
        free(q); free(tmp);
    }
    return 0;
}
#endif

extern "C" int h264e_hip_pool_create(h264e_hip_pool_t **pool, int device, int width, int height, int nchains,
                                     int frames_resident, int slots)
{
    if (!pool || width <= 0 || height <= 0 || ((width | height) & 1) || nchains <= 0 || frames_resident <= 0 || slots <= 0)
        FAIL("h264e_hip_pool_create: bad argument");
    h264e_hip_pool_t *p = (h264e_hip_pool_t *)calloc(1, sizeof(*p));
    if (!p) FAIL("out of host memory");
    p->device = device; p->nchains = nchains; p->frames_resident = frames_resident; p->slots = slots;
    h264e_geom_t &G = p->G;
    G.width = width; G.height = height;
    G.nmbx = (width + 15) >> 4; G.nmby = (height + 15) >> 4; G.nmb = G.nmbx*G.nmby;
    G.W = G.nmbx*16; G.H = G.nmby*16;
    G.cropping = !!((width | height) & 15);
    G.lim_x0 = G.lim_y0 = -14*4;                                    /* h264-lab.h:6322-6324, MV_GUARD 14 */
    G.lim_x1 = (G.W - 2)*4; G.lim_y1 = (G.H - 2)*4;
    G.row_words = G.nmbx*(H264E_ROW_BYTES_PER_MB/4);
    /* knobs for the failure-path tests only: a tiny row bit buffer (overflow), a short spin bound, a row that never publishes, an
     * asynchronous upload that fails.  They are looked at ONLY under the explicit switch H264E_TEST_KNOBS=1, so that a stray
     * H264E_TEST_* variable inherited from somebody's environment cannot make a production encode fail. */
    const int knobs = getenv("H264E_TEST_KNOBS") && atoi(getenv("H264E_TEST_KNOBS")) == 1;
    G.spin_limit = 1u << 24;
    G.test_stall_row = -1;
    p->test_upload_fail_at = -1;
    if (knobs)
    {
        if (getenv("H264E_TEST_ROW_BYTES_PER_MB")) { const int b = atoi(getenv("H264E_TEST_ROW_BYTES_PER_MB"))/4; G.row_words = G.nmbx*(b > 1 ? b : 1); }
        if (getenv("H264E_TEST_SPIN_LIMIT")) G.spin_limit = (unsigned)atol(getenv("H264E_TEST_SPIN_LIMIT"));
        if (getenv("H264E_TEST_STALL_ROW")) G.test_stall_row = atoi(getenv("H264E_TEST_STALL_ROW"));
        if (getenv("H264E_TEST_UPLOAD_FAIL_AT")) p->test_upload_fail_at = atoi(getenv("H264E_TEST_UPLOAD_FAIL_AT"));
    }
    p->frame_bytes = (size_t)width*height*3/2;
    p->waves = getenv("H264E_WAVES") ? atoi(getenv("H264E_WAVES")) : 0;                  /* 1 / 2: forced (A-B measurements); else chosen per launch */
    if (p->waves != 1 && p->waves != 2 && p->waves != 4) p->waves = 0;
#ifndef H264E_EMU
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    {
        free(p);
        FAIL("no HIP device: the HIP path is mandatory (there is no CPU fallback)");
    }
    if (hipSetDevice(device) != hipSuccess) { free(p); FAIL("hipSetDevice(%d) failed", device); }
    {
        const int share = getenv("H264E_SHARE_DEVICE") && atoi(getenv("H264E_SHARE_DEVICE")) == 1;
        if (process_guard_acquire(device)) { free(p); return -1; }
        p->guarded = !share;
    }
    if (hipStreamCreate(&p->stream) != hipSuccess || hipStreamCreate(&p->copy_stream) != hipSuccess || hipStreamCreate(&p->abort_stream) != hipSuccess) { if (p->guarded) process_guard_release(device); free(p); FAIL("hipStreamCreate failed"); }
    for (int i = 0; i < TASK_RING; i++) for (int k = 0; k < 3; k++) (void)hipEventCreate(&p->ev[i][k]);
    (void)hipEventCreate(&p->ev_t0); (void)hipEventCreate(&p->ev_t1); (void)hipEventCreate(&p->ev_prep);
#endif
    p->chains_host = (h264e_chain_dev_t *)calloc((size_t)nchains, sizeof(h264e_chain_dev_t));
    p->clu_dev = (int32_t **)calloc((size_t)nchains, sizeof(int32_t *));
    p->ref_sel = (int *)calloc((size_t)nchains, sizeof(int));
    p->traj_dev = (int32_t **)calloc((size_t)nchains, sizeof(int32_t *));
    p->traj_cur = (int *)calloc((size_t)nchains, sizeof(int));
    p->slot_launch = (int *)calloc((size_t)nchains, sizeof(int));
    p->host_rbsp = (uint8_t **)calloc((size_t)nchains, sizeof(uint8_t *));
    p->host_mbrec = (h264e_hip_mbrec_t **)calloc((size_t)nchains, sizeof(h264e_hip_mbrec_t *));
    int bad = 0;
    const size_t plane = (size_t)G.W*G.H*3/2;
    const uint32_t arena_cap = (uint32_t)((size_t)slots*((size_t)G.nmb*640 + 1024));
    /* host-mapped mirror per slot: sized for ordinary frames (160 B per macroblock; a 1080p key frame at QP 26 needs ~20); a
     * frame that does not fit stays in the slot's device NAL arena (worst-case size) and is fetched with a copy */
    const uint32_t nal_cap = (uint32_t)((size_t)G.nmb*660 + 4096);
    p->host_rbsp_cap = getenv("H264E_HOST_MIRROR_BYTES") ? (uint32_t)atol(getenv("H264E_HOST_MIRROR_BYTES")) : (uint32_t)((size_t)G.nmb*160 + 65536);
    if (p->host_rbsp_cap > nal_cap) p->host_rbsp_cap = nal_cap;
    /* One device allocation and one host-mapped allocation per pool, carved by a bump pointer: pass 0 sizes them, pass 1
     * hands out the pointers.  (Hundreds of separate small allocations get small page-table fragments; one large block is
     * mapped with large ones, and every macroblock touches about ten of these buffers.) */
    for (int pass = 0; pass < 2 && !bad; pass++)
    {
        size_t pos = 0, hpos = 0;
        uint8_t *base = pass ? p->heap : 0, *hbase = pass ? p->hheap : 0;
        auto carve = [&](size_t n, size_t align) -> void * { pos = (pos + align - 1) & ~(align - 1); void *r = base ? base + pos : 0; pos += n ? n : 1; return r; };
        auto hcarve = [&](size_t n) -> void * { hpos = (hpos + 255) & ~(size_t)255; void *r = hbase ? hbase + hpos : 0; hpos += n; return r; };
        p->clip = (uint8_t *)carve(p->frame_bytes*(size_t)frames_resident, 4096);
        p->chains_dev = (h264e_chain_dev_t *)carve(sizeof(h264e_chain_dev_t)*(size_t)nchains, 256);
        p->tasks_dev = (h264e_frame_task_t *)carve(sizeof(h264e_frame_task_t)*(size_t)nchains*TASK_RING, 256);
        p->progress_all = (int *)carve(sizeof(int)*(size_t)nchains*G.nmby, 256);
        p->errflag = (int *)carve(sizeof(int), 256);
        p->stepflags = (int *)carve(sizeof(int)*2*(size_t)nchains, 256);
        p->abort_dev = (int *)carve(64, 256);
        p->walkrec = (h264e_walkrec_t *)carve(sizeof(h264e_walkrec_t)*(size_t)nchains, 256);
        p->ssd_dev = (unsigned long long *)carve(sizeof(unsigned long long)*3*(size_t)nchains, 256);
        p->order = (uint32_t *)carve(sizeof(uint32_t)*(size_t)nchains*(G.nmby + 1), 256);
        p->host_done = (h264e_hostdone_t *)hcarve(sizeof(h264e_hostdone_t)*(size_t)nchains);
        p->abort_word = (int *)hcarve(64);
        for (int c = 0; c < nchains; c++)
        {
            h264e_chain_dev_t &C = p->chains_host[c];
            uint8_t *rec = (uint8_t *)carve(2*plane, 4096);
            for (int k = 0; k < 2; k++)
            {
                C.rec[k][0] = rec + k*plane;
                C.rec[k][1] = C.rec[k][0] + (size_t)G.W*G.H;
                C.rec[k][2] = C.rec[k][1] + (size_t)G.W*G.H/4;
            }
            C.bottom = (h264e_mbbottom_t *)carve(sizeof(h264e_mbbottom_t)*(size_t)G.nmb, 256);
            C.pend = (h264e_mbpend_t *)carve(sizeof(h264e_mbpend_t)*(size_t)G.nmb, 256);
            C.progress = p->progress_all + (size_t)c*G.nmby;
            C.rowbits = (uint32_t *)carve(sizeof(uint32_t)*(size_t)G.nmby*G.row_words, 256);
            C.rowmeta = (h264e_rowmeta_t *)carve(sizeof(h264e_rowmeta_t)*(size_t)G.nmby, 256);
            C.mbrec = (h264e_mbrec_t *)carve(sizeof(h264e_mbrec_t)*(size_t)G.nmb*slots, 256);
            C.arena = (uint8_t *)carve(arena_cap, 256);
            C.arena_cap = arena_cap;
            C.nal_arena = slots == 1 ? (uint8_t *)carve(nal_cap, 256) : 0;
            C.nal_cap = slots == 1 ? nal_cap : 0;
            C.cursor = (uint32_t *)carve(16, 256);
            C.fout = (h264e_frameout_t *)carve(sizeof(h264e_frameout_t)*(size_t)slots, 256);
            C.prof = (unsigned long long *)carve(sizeof(unsigned long long)*32, 256);
            C.far_reads = (int *)carve(16, 256);
            p->clu_dev[c] = (int32_t *)carve(sizeof(int32_t)*2*(size_t)G.nmb, 256);      /* per-macroblock mv_clusters array of a re-encode */
            p->traj_dev[c] = slots == 1 ? (int32_t *)carve(sizeof(int32_t)*4*(size_t)G.nmb, 256) : 0;   /* two walk trajectories (device-side validation) */
            if (slots == 1)        /* streaming pools keep one result per chain slot: give each a host-mapped mirror */
            {
                p->host_rbsp[c] = (uint8_t *)hcarve(p->host_rbsp_cap + 64);
                p->host_mbrec[c] = (h264e_hip_mbrec_t *)hcarve(sizeof(h264e_hip_mbrec_t)*(size_t)G.nmb + 64);
            }
        }
        if (!pass)
        {
            p->heap_bytes = pos + 4096; p->hheap_bytes = hpos + 4096;
            bad |= dev_malloc((void **)&p->heap, p->heap_bytes);
            bad |= host_malloc((void **)&p->hheap, p->hheap_bytes);
        }
    }
    if (bad)
    {
        h264e_hip_pool_destroy(p);
        FAIL("device allocation failed");
    }
#ifdef H264E_EMU
    memcpy(p->chains_dev, p->chains_host, sizeof(h264e_chain_dev_t)*(size_t)nchains);
#else
    (void)hipMemset(p->heap, 0, p->heap_bytes);
    if (hipMemcpy(p->chains_dev, p->chains_host, sizeof(h264e_chain_dev_t)*(size_t)nchains, hipMemcpyHostToDevice) != hipSuccess)
    {
        h264e_hip_pool_destroy(p);
        FAIL("descriptor upload failed");
    }
    p->order_host = (uint32_t *)malloc(sizeof(uint32_t)*(size_t)nchains*(G.nmby + 1));
    if (!p->order_host) { h264e_hip_pool_destroy(p); FAIL("out of host memory"); }
    p->order_jobs = -1;
#endif
    *pool = p;
    return 0;
}

extern "C" int h264e_hip_upload_i420(h264e_hip_pool_t *p, int first, int nframes, const uint8_t *host)
{
    if (!p || first < 0 || nframes < 0 || first + nframes > p->frames_resident) FAIL("upload_i420: bad range");
#ifdef H264E_EMU
    memcpy(p->clip + p->frame_bytes*(size_t)first, host, p->frame_bytes*(size_t)nframes);
#else
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpyAsync(p->clip + p->frame_bytes*(size_t)first, host, p->frame_bytes*(size_t)nframes, hipMemcpyHostToDevice, p->stream));
#endif
    return 0;
}

extern "C" int h264e_hip_upload_i420_async(h264e_hip_pool_t *p, int first, int nframes, const uint8_t *host)
{
    if (!p || first < 0 || nframes < 0 || first + nframes > p->frames_resident) FAIL("upload_i420_async: bad range");
    if (p->async_uploads++ == p->test_upload_fail_at) FAIL("upload_i420_async: injected failure (H264E_TEST_UPLOAD_FAIL_AT)");
#ifdef H264E_EMU
    memcpy(p->clip + p->frame_bytes*(size_t)first, host, p->frame_bytes*(size_t)nframes);
#else
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpyAsync(p->clip + p->frame_bytes*(size_t)first, host, p->frame_bytes*(size_t)nframes, hipMemcpyHostToDevice, p->copy_stream));
#endif
    return 0;
}

extern "C" int h264e_hip_upload_wait(h264e_hip_pool_t *p)
{
    if (!p) FAIL("upload_wait: null pool");
#ifndef H264E_EMU
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->copy_stream));
#endif
    return 0;
}

extern "C" int h264e_hip_upload_busy(h264e_hip_pool_t *p)
{
#ifdef H264E_EMU
    (void)p;
    return 0;
#else
    if (!p) return 0;
    (void)hipSetDevice(p->device);
    const hipError_t e = hipStreamQuery(p->copy_stream);
    if (e == hipErrorNotReady) return 1;
    if (e != hipSuccess) FAIL("upload: %s", hipGetErrorString(e));      /* -1: the copy was lost, not finished */
    return 0;
#endif
}

extern "C" void *h264e_hip_host_alloc(size_t bytes)
{
#ifdef H264E_EMU
    return malloc(bytes ? bytes : 1);
#else
    void *q = 0;
    return hipHostMalloc(&q, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? q : 0;
#endif
}

extern "C" void h264e_hip_host_free(void *q)
{
#ifdef H264E_EMU
    free(q);
#else
    if (q) (void)hipHostFree(q);
#endif
}

#ifndef H264E_EMU
/* sum of squared differences of one plane pair; grid.y = frame, grid.z = plane, grid.x strides over the samples */
__global__ void h264e_ssd_kernel(const uint8_t *clip, size_t frame_bytes, int width, int height, int in0, int in_mod,
                                 const h264e_chain_dev_t *chains, int pic0, int pic_mod, int W, unsigned long long *out)
{
    const int i = (int)blockIdx.y, pl = (int)blockIdx.z;
    const int w = width >> (pl ? 1 : 0), h = height >> (pl ? 1 : 0), ps = W >> (pl ? 1 : 0);
    const uint8_t *a = clip + frame_bytes*(size_t)((in0 + i) % in_mod) + (pl ? (size_t)width*height + (pl == 2 ? (size_t)(width/2)*(height/2) : 0) : 0);
    const uint8_t *b = chains[(pic0 + i) % pic_mod].rec[0][pl];
    unsigned long long s = 0;
    for (int k = (int)(blockIdx.x*blockDim.x + threadIdx.x); k < w*h; k += (int)(gridDim.x*blockDim.x))
    {
        const int y = k / w, x = k - y*w, d = (int)a[k] - (int)b[(size_t)y*ps + x];
        s += (unsigned long long)(d*d);
    }
    for (int o = 32; o; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(out + 3*i + pl, s);
}
#endif

extern "C" int h264e_hip_ssd_frames(h264e_hip_pool_t *p, int n, int in0, int in_mod, int pic0, int pic_mod, uint64_t *out)
{
    if (!p || !out || n <= 0 || n > p->nchains || in_mod <= 0 || in_mod > p->frames_resident || pic_mod <= 0 || pic_mod > p->nchains) FAIL("ssd_frames: bad argument");
    const h264e_geom_t &G = p->G;
#ifdef H264E_EMU
    for (int i = 0; i < n; i++)
        for (int pl = 0; pl < 3; pl++)
        {
            const int w = G.width >> (pl ? 1 : 0), h = G.height >> (pl ? 1 : 0), ps = G.W >> (pl ? 1 : 0);
            const uint8_t *a = p->clip + p->frame_bytes*(size_t)((in0 + i) % in_mod) + (pl ? (size_t)G.width*G.height + (pl == 2 ? (size_t)(G.width/2)*(G.height/2) : 0) : 0);
            const uint8_t *b = p->chains_host[(pic0 + i) % pic_mod].rec[0][pl];
            uint64_t s = 0;
            for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) { const int d = (int)a[(size_t)y*w + x] - (int)b[(size_t)y*ps + x]; s += (uint64_t)(d*d); }
            out[3*i + pl] = s;
        }
#else
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemsetAsync(p->ssd_dev, 0, sizeof(unsigned long long)*3*(size_t)n, p->stream));
    hipLaunchKernelGGL(h264e_ssd_kernel, dim3(64, (unsigned)n, 3), dim3(256), 0, p->stream, (const uint8_t *)p->clip, p->frame_bytes, G.width, G.height,
                       in0, in_mod, (const h264e_chain_dev_t *)p->chains_dev, pic0, pic_mod, G.W, p->ssd_dev);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, p->ssd_dev, sizeof(unsigned long long)*3*(size_t)n, hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
#endif
    return 0;
}

extern "C" int h264e_hip_read_recon_slot(h264e_hip_pool_t *p, int slot, uint8_t *dst)
{
    if (!p || !dst || slot < 0 || slot >= p->nchains) FAIL("read_recon_slot: bad argument");
    const size_t n = (size_t)p->G.W*p->G.H*3/2;
#ifdef H264E_EMU
    memcpy(dst, p->chains_host[slot].rec[0][0], n);
#else
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpy(dst, p->chains_host[slot].rec[0][0], n, hipMemcpyDeviceToHost));
#endif
    return 0;
}

extern "C" int h264e_hip_upload_planes(h264e_hip_pool_t *p, int index, const uint8_t *const yuv[3], const int stride[3])
{
    if (!p || index < 0 || index >= p->frames_resident) FAIL("upload_planes: bad index");
    uint8_t *d = p->clip + p->frame_bytes*(size_t)index;
    for (int c = 0; c < 3; c++)
    {
        const int w = p->G.width >> (c ? 1 : 0), h = p->G.height >> (c ? 1 : 0);
#ifdef H264E_EMU
        for (int y = 0; y < h; y++) memcpy(d + (size_t)y*w, yuv[c] + (size_t)y*stride[c], (size_t)w);
#else
        HIPCHK(hipSetDevice(p->device));
        HIPCHK(hipMemcpy2DAsync(d, (size_t)w, yuv[c], (size_t)stride[c], (size_t)w, (size_t)h, hipMemcpyHostToDevice, p->stream));
#endif
        d += (size_t)w*h;
    }
    return 0;
}

extern "C" int h264e_hip_generate_synth(h264e_hip_pool_t *p, int first, int nframes, int t0, uint32_t seed)
{
    if (!p || first < 0 || nframes < 0 || first + nframes > p->frames_resident) FAIL("generate_synth: bad range");
    for (int i = 0; i < nframes; i++)
    {
        uint8_t *d = p->clip + p->frame_bytes*(size_t)(first + i);
#ifdef H264E_EMU
        for (int k = 0; k < (int)p->frame_bytes; k++) d[k] = sv_sample(p->G.width, p->G.height, t0 + i, seed, k);
#else
        HIPCHK(hipSetDevice(p->device));
        hipLaunchKernelGGL(h264e_synth_kernel, dim3(1024), dim3(256), 0, p->stream, d, p->G.width, p->G.height, t0 + i, seed);
#endif
    }
#ifndef H264E_EMU
    HIPCHK(hipGetLastError());
#endif
    return 0;
}

extern "C" int h264e_hip_sync(h264e_hip_pool_t *p)
{
    if (!p) FAIL("sync: null pool");
#ifndef H264E_EMU
    HIPCHK(hipSetDevice(p->device));
    if (p->group)
    {
        /* the merged launch of the round this pool submitted in (all members' jobs) */
        const hipError_t eg = hipEventSynchronize(p->group->ev_done);
        if (eg != hipSuccess) FAIL("group launch: %s", hipGetErrorString(eg));
        if (p->profile && p->group->nmembers && p->group->member[0] == p)
        {
            float a = 0;
            if (hipEventElapsedTime(&a, p->group->ev_t0, p->group->ev_t1) == hipSuccess) { p->prof_mb_ms += a; p->prof_launches++; }
        }
    }
    {
        const hipError_t es = hipStreamSynchronize(p->stream);
        device_release(p);              /* drained (or lost): the next launch on this device may go */
        if (es != hipSuccess) FAIL("hipStreamSynchronize: %s", hipGetErrorString(es));
    }
    for (int i = 0; i < p->ev_pending; i++)
    {
        float a = 0, b = 0;
        HIPCHK(hipEventElapsedTime(&a, p->ev[i][0], p->ev[i][1]));
        HIPCHK(hipEventElapsedTime(&b, p->ev[i][1], p->ev[i][2]));
        p->prof_mb_ms += a; p->prof_splice_ms += b; p->prof_launches++;
    }
    p->ev_pending = 0;
    int err = 0;
    HIPCHK(hipMemcpy(&err, p->errflag, sizeof(int), hipMemcpyDeviceToHost));
    if (err)
    {
        (void)hipMemset(p->errflag, 0, sizeof(int));
        FAIL("macroblock kernel gave up waiting for the row above (bounded spin expired)");
    }
#endif
    device_release(p);
    p->pending = 0;
    return 0;
}

/* give the device back after a failure in the middle of a launch sequence (no error reporting of its own) */
extern "C" void h264e_hip_release(h264e_hip_pool_t *p)
{
    if (!p) return;
#ifndef H264E_EMU
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
#endif
    device_release(p);
}

#ifndef H264E_EMU
/* variant: 0 = intra frames only (one wave per row, 4 per SIMD), 1 = one wave per row, 2 = two waves per row (3 per SIMD), 4 = two waves
 * per row at 4 per SIMD */
static void launch_mb_kernel(const h264e_geom_t &G, int narrow, int variant, unsigned nblocks, const h264e_frame_task_t *td, const uint32_t *od, hipStream_t st)
{
    const dim3 grid(nblocks);
    if (variant == 0) hipLaunchKernelGGL((h264e_mb_kernel<GEOM_INTRA, 1, H264E_WPEI>), grid, dim3(64), 0, st, G, td, od);
    else if (variant == 4)
    {
        if (narrow) hipLaunchKernelGGL((h264e_mb_kernel<GEOM_NARROW, 2, 4>), grid, dim3(128), 0, st, G, td, od);
        else hipLaunchKernelGGL((h264e_mb_kernel<GEOM_WIDE, 2, 4>), grid, dim3(128), 0, st, G, td, od);
    } else if (variant == 2)
    {
        if (narrow) hipLaunchKernelGGL((h264e_mb_kernel<GEOM_NARROW, 2, H264E_WPE2>), grid, dim3(128), 0, st, G, td, od);
        else hipLaunchKernelGGL((h264e_mb_kernel<GEOM_WIDE, 2, H264E_WPE2>), grid, dim3(128), 0, st, G, td, od);
    } else
    {
        if (narrow) hipLaunchKernelGGL((h264e_mb_kernel<GEOM_NARROW, 1, H264E_WPE1>), grid, dim3(64), 0, st, G, td, od);
        else hipLaunchKernelGGL((h264e_mb_kernel<GEOM_WIDE, 1, H264E_WPE1>), grid, dim3(64), 0, st, G, td, od);
    }
}
#endif

/* ------------------------------------------------------------------ launch groups: several streams in ONE launch
 *
 * A single-slice stream is latency bound: after every mis-speculated mv_clusters state its pipeline drains and refills, and the chip
 * idles meanwhile.  Independent streams of the same picture size can fill each other's gaps -- but not as separate launches (see
 * g_device_lock: two persistent launches side by side can starve each other).  A group merges the launches of its member pools into ONE
 * grid: every member submits as usual (h264e_hip_submit blocks until all members that are still encoding have submitted or left), the
 * last one to arrive concatenates the jobs, interleaves the members' dispatch orders by start step -- so the streams advance in lock step
 * and every workgroup still only waits for workgroups in front of it -- and launches once.  Each job keeps its own pool's buffers, abort
 * word, error word and host mirrors (h264e_frame_task_t), so one stream's abort stops only its own jobs; h264e_hip_sync of a member
 * returns when the merged launch has drained.  Members are encoded by different host threads (H264E_clip_encode_multi).
 */
extern "C" int h264e_hip_group_create(h264e_hip_group_t **out, int device)
{
    if (!out) FAIL("group_create: null argument");
    h264e_hip_group_t *g = (h264e_hip_group_t *)calloc(1, sizeof(*g));
    if (!g) FAIL("out of host memory");
    g->device = device;
#ifndef H264E_EMU
    pthread_mutex_init(&g->mu, 0); pthread_cond_init(&g->cv, 0);
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&g->stream) != hipSuccess) { free(g); FAIL("group_create: no stream on device %d", device); }
    (void)hipEventCreate(&g->ev_done); (void)hipEventCreate(&g->ev_t0); (void)hipEventCreate(&g->ev_t1);
#endif
    *out = g;
    return 0;
}

#ifndef H264E_EMU
/* all members that are still in the group have submitted: merge and launch (g->mu held).  Members whose launches differ in kernel
 * variant (window geometry, waves per row) go in separate launches, one after the other. */
static int group_launch_locked(h264e_hip_group_t *g)
{
    int rc = 0;
    if (hipSetDevice(g->device) != hipSuccess) rc = -1;
    for (int variant = 0; variant < 10 && !rc; variant++)
    {
        const int narrow = variant & 1, waves = variant >> 1;
        int idx[H264E_GROUP_MAX], n = 0, jobs = 0;
        for (int k = 0; k < g->nmembers; k++)
            if (g->pend[k] && g->pend_narrow[k] == narrow && g->pend_waves[k] == waves) { idx[n++] = k; jobs += g->pend_jobs[k]; }
        if (!n) continue;
        const h264e_geom_t &G = g->member[idx[0]]->G;
        const int rows = G.nmby + 1, lag = narrow ? H264E_NARROW_FRAME_LAG : H264E_FRAME_LAG;
        const size_t total = (size_t)jobs*rows;
        if (jobs >= 65536) { snprintf(g_err, sizeof(g_err), "group launch: too many jobs"); rc = -1; break; }
        if ((size_t)jobs > g->tasks_cap)
        {
            if (g->tasks_dev) (void)hipFree(g->tasks_dev);
            g->tasks_cap = (size_t)jobs + 64;
            if (hipMalloc((void **)&g->tasks_dev, sizeof(h264e_frame_task_t)*g->tasks_cap) != hipSuccess) { g->tasks_dev = 0; g->tasks_cap = 0; snprintf(g_err, sizeof(g_err), "group launch: device allocation failed"); rc = -1; break; }
        }
        if (total > g->order_cap)
        {
            if (g->order_dev) (void)hipFree(g->order_dev);
            g->order_cap = total + 4096;
            if (hipMalloc((void **)&g->order_dev, sizeof(uint32_t)*g->order_cap) != hipSuccess) { g->order_dev = 0; g->order_cap = 0; snprintf(g_err, sizeof(g_err), "group launch: device allocation failed"); rc = -1; break; }
        }
        h264e_frame_task_t *th = (h264e_frame_task_t *)malloc(sizeof(h264e_frame_task_t)*(size_t)jobs);
        uint32_t *oh = (uint32_t *)malloc(sizeof(uint32_t)*total);
        if (!th || !oh) { free(th); free(oh); snprintf(g_err, sizeof(g_err), "out of host memory"); rc = -1; break; }
        /* jobs member after member; dispatch order by start step lag*job + 2*row (a counting sort over all members: ties go member by
         * member, so the streams are interleaved frame by frame) */
        int base[H264E_GROUP_MAX], maxjobs = 0;
        for (int i = 0, b = 0; i < n; i++)
        {
            base[i] = b;
            memcpy(th + b, g->pend_tasks[idx[i]], sizeof(h264e_frame_task_t)*(size_t)g->pend_jobs[idx[i]]);
            b += g->pend_jobs[idx[i]];
            if (g->pend_jobs[idx[i]] > maxjobs) maxjobs = g->pend_jobs[idx[i]];
        }
        const int maxkey = lag*(maxjobs - 1) + 2*(rows - 1);
        int *start = (int *)calloc((size_t)maxkey + 2, sizeof(int));
        if (!start) { free(th); free(oh); snprintf(g_err, sizeof(g_err), "out of host memory"); rc = -1; break; }
        for (int i = 0; i < n; i++) for (int j = 0; j < g->pend_jobs[idx[i]]; j++) for (int r = 0; r < rows; r++) start[lag*j + 2*r + 1]++;
        for (int k = 0; k <= maxkey; k++) start[k + 1] += start[k];
        for (int j = 0; j < maxjobs; j++)
            for (int i = 0; i < n; i++)
                if (j < g->pend_jobs[idx[i]])
                    for (int r = 0; r < rows; r++) oh[start[lag*j + 2*r]++] = ((uint32_t)(base[i] + j) << 16) | (uint32_t)r;
        free(start);
        /* the members prepared their slots (progress counters, ...) on their own streams: the launch waits for all of that */
        for (int i = 0; i < n && !rc; i++)
        {
            h264e_hip_pool_t *p = g->member[idx[i]];
            if (hipEventRecord(p->ev_prep, p->stream) != hipSuccess || hipStreamWaitEvent(g->stream, p->ev_prep, 0) != hipSuccess) rc = -1;
        }
        if (!rc && (hipMemcpyAsync(g->tasks_dev, th, sizeof(h264e_frame_task_t)*(size_t)jobs, hipMemcpyHostToDevice, g->stream) != hipSuccess ||
                    hipMemcpyAsync(g->order_dev, oh, sizeof(uint32_t)*total, hipMemcpyHostToDevice, g->stream) != hipSuccess)) rc = -1;
        free(th); free(oh);            /* pageable sources: staged before the calls return */
        if (!rc)
        {
            (void)hipEventRecord(g->ev_t0, g->stream);
            launch_mb_kernel(G, narrow, waves, (unsigned)total, g->tasks_dev, g->order_dev, g->stream);
            (void)hipEventRecord(g->ev_t1, g->stream);
            if (hipGetLastError() != hipSuccess) rc = -1;
        }
        if (rc && !g_err[0]) snprintf(g_err, sizeof(g_err), "group launch failed");
    }
    if (hipEventRecord(g->ev_done, g->stream) != hipSuccess) rc = -1;
    for (int k = 0; k < g->nmembers; k++) { free(g->pend_tasks[k]); g->pend_tasks[k] = 0; g->pend[k] = 0; }
    g->arrived = 0;
    g->failed = rc;
    g->round++;
    pthread_cond_broadcast(&g->cv);
    return rc;
}

/* a member's launch: hand it to the group and wait until the merged launch is on its way */
static int group_submit(h264e_hip_pool_t *p, const h264e_frame_task_t *host, int njobs, int narrow, int waves)
{
    h264e_hip_group_t *g = p->group;
    int rc = 0, k;
    pthread_mutex_lock(&g->mu);
    for (k = 0; k < g->nmembers && g->member[k] != p; k++) ;
    if (k == g->nmembers) { pthread_mutex_unlock(&g->mu); FAIL("group_submit: not a member"); }
    g->pend_tasks[k] = (h264e_frame_task_t *)malloc(sizeof(h264e_frame_task_t)*(size_t)njobs);
    if (!g->pend_tasks[k]) { pthread_mutex_unlock(&g->mu); FAIL("out of host memory"); }
    memcpy(g->pend_tasks[k], host, sizeof(h264e_frame_task_t)*(size_t)njobs);
    g->pend_jobs[k] = njobs; g->pend_narrow[k] = narrow; g->pend_waves[k] = waves; g->pend[k] = 1;
    p->group_round = g->round;
    g->arrived++;
    if (g->arrived == g->nmembers) rc = group_launch_locked(g);
    else
    {
        const int r = g->round;
        while (g->round == r) pthread_cond_wait(&g->cv, &g->mu);
        rc = g->failed;
    }
    pthread_mutex_unlock(&g->mu);
    if (rc && !g_err[0]) snprintf(g_err, sizeof(g_err), "group launch failed");
    return rc;
}
#endif

extern "C" int h264e_hip_group_join(h264e_hip_group_t *g, h264e_hip_pool_t *p)
{
    if (!g || !p || p->group) FAIL("group_join: bad argument");
#ifndef H264E_EMU
    pthread_mutex_lock(&g->mu);
    int bad = g->nmembers >= H264E_GROUP_MAX || p->device != g->device || g->arrived;
    if (!bad && g->nmembers)
    {
        const h264e_geom_t &A = g->member[0]->G, &B = p->G;
        bad = A.width != B.width || A.height != B.height || A.row_words != B.row_words || A.spin_limit != B.spin_limit;
    }
    if (!bad) { g->member[g->nmembers++] = p; p->group = g; }
    pthread_mutex_unlock(&g->mu);
    if (bad) FAIL("group_join: the group is full, busy, on another device or holds another picture size");
#else
    p->group = g; g->member[g->nmembers++] = p;     /* the emulation runs every submit by itself, at once */
#endif
    return 0;
}

extern "C" void h264e_hip_group_leave(h264e_hip_group_t *g, h264e_hip_pool_t *p)
{
    if (!g || !p || p->group != g) return;
#ifndef H264E_EMU
    pthread_mutex_lock(&g->mu);
    int k;
    for (k = 0; k < g->nmembers && g->member[k] != p; k++) ;
    if (k < g->nmembers)
    {
        if (g->pend[k]) { free(g->pend_tasks[k]); g->arrived--; }
        for (; k + 1 < g->nmembers; k++)
        {
            g->member[k] = g->member[k + 1]; g->pend_tasks[k] = g->pend_tasks[k + 1]; g->pend_jobs[k] = g->pend_jobs[k + 1];
            g->pend_narrow[k] = g->pend_narrow[k + 1]; g->pend_waves[k] = g->pend_waves[k + 1]; g->pend[k] = g->pend[k + 1];
        }
        g->nmembers--;
        g->pend_tasks[g->nmembers] = 0; g->pend[g->nmembers] = 0;
        /* the others may have been waiting for this member only */
        if (g->nmembers && g->arrived == g->nmembers) (void)group_launch_locked(g);
    }
    p->group = 0;
    pthread_mutex_unlock(&g->mu);
#else
    for (int k = 0; k < g->nmembers; k++) if (g->member[k] == p) { g->member[k] = g->member[--g->nmembers]; break; }
    p->group = 0;
#endif
}

extern "C" void h264e_hip_group_destroy(h264e_hip_group_t *g)
{
    if (!g) return;
#ifndef H264E_EMU
    (void)hipSetDevice(g->device);
    (void)hipStreamSynchronize(g->stream);
    for (int k = 0; k < g->nmembers; k++) { g->member[k]->group = 0; free(g->pend_tasks[k]); }
    if (g->tasks_dev) (void)hipFree(g->tasks_dev);
    if (g->order_dev) (void)hipFree(g->order_dev);
    (void)hipEventDestroy(g->ev_done); (void)hipEventDestroy(g->ev_t0); (void)hipEventDestroy(g->ev_t1);
    (void)hipStreamDestroy(g->stream);
    pthread_mutex_destroy(&g->mu); pthread_cond_destroy(&g->cv);
#else
    for (int k = 0; k < g->nmembers; k++) g->member[k]->group = 0;
#endif
    free(g);
}

extern "C" int h264e_hip_submit(h264e_hip_pool_t *p, const h264e_hip_task_t *tasks)
{
    if (!p || !tasks) FAIL("submit: null argument");
    const h264e_geom_t &G = p->G;
    if (p->pending >= TASK_RING - 1 && h264e_hip_sync(p)) return -1;
    h264e_frame_task_t *host = (h264e_frame_task_t *)calloc((size_t)p->nchains, sizeof(h264e_frame_task_t));
    if (!host) FAIL("out of host memory");
    int any = 0, any_narrow = 0, any_wide = 0, njobs = 0, all_intra = 1, max_slices = 1;
    const int launch_id = ++p->launch_counter;
    for (int c = 0; c < p->nchains; c++)
    {
        const h264e_hip_task_t &t = tasks[c];
        h264e_frame_task_t &d = host[c];
        d.active = t.active;
        if (!t.active) continue;
        if (t.frame_index < 0 || t.frame_index >= p->frames_resident || t.frame_slot < 0 || t.frame_slot >= p->slots ||
            t.qp < 10 || t.qp > 51 || t.hdr_nbits < 0 || t.hdr_nbits > 56 || t.nslices < 0 || t.nslices > H264E_MAX_SLICES || t.nslices > G.nmby)
        {
            free(host);
            FAIL("submit: bad task for chain %d", c);
        }
        any = 1; njobs = c + 1;
        if (t.slice_type != 2) all_intra = 0;
        if (t.nslices > max_slices) max_slices = t.nslices;
        const uint8_t *f = p->clip + p->frame_bytes*(size_t)t.frame_index;
        d.in[0] = f; d.in[1] = f + (size_t)G.width*G.height; d.in[2] = d.in[1] + (size_t)(G.width/2)*(G.height/2);
        d.in_stride[0] = G.width; d.in_stride[1] = d.in_stride[2] = G.width/2;
        d.slice_type = t.slice_type; d.qp = t.qp; d.speed = t.speed;
        d.no_deblock = (t.speed == 8 || t.speed == 10);                 /* h264-lab.h:6717 */
        if (t.stream_mode)
        {
            /* temporal wavefront: job c builds the picture of chain slot t.slot from the picture of slot t.ref_slot */
            if (t.slot < 0 || t.slot >= p->nchains || t.ref_slot >= p->nchains || (t.slice_type == 0 && t.ref_slot < 0) ||
                (t.ref_in_flight && t.ref_slot < 0))
            {
                free(host);
                FAIL("submit: bad stream task %d", c);
            }
            d.chain = t.slot;
            d.arena_reset = 1;
            if (p->host_rbsp[t.slot] && p->host_mbrec[t.slot])
            {
                d.host_done = p->host_done + t.slot;
                d.host_rbsp = p->host_rbsp[t.slot]; d.host_rbsp_cap = p->host_rbsp_cap;
                d.host_mbrec = (h264e_mbrec_t *)p->host_mbrec[t.slot];
                d.abort_word = p->abort_dev;
                p->host_done[t.slot].done = 0;
                p->slot_launch[t.slot] = launch_id;
                if (t.walk_on_device && p->traj_dev[t.slot])
                {
                    d.walk_on_device = 1;
                    d.walk_quiet = t.walk_quiet;
                    d.exact_state[0] = t.exact_state[0]; d.exact_state[1] = t.exact_state[1];
                    d.walk_out = p->walkrec + t.slot;
                    {
                        const int par = t.walk_parent > 0 ? t.walk_parent - 1 : c - 1;
                        d.walk_prev = (par >= 0 && par < c && tasks[par].active && tasks[par].stream_mode && tasks[par].walk_on_device) ? p->walkrec + tasks[par].slot : 0;
                    }
                    d.traj_out = p->traj_dev[t.slot] + (size_t)(p->traj_cur[t.slot] ^ 1)*2*G.nmb;
                }
            }
            for (int k = 0; k < 3; k++)
            {
                d.dec[k] = p->chains_host[t.slot].rec[0][k];
                d.ref[k] = t.ref_slot >= 0 ? p->chains_host[t.ref_slot].rec[0][k] : p->chains_host[t.slot].rec[1][k];
            }
            d.dep_progress = t.ref_in_flight ? p->chains_host[t.ref_slot].progress : 0;
        } else
        {
            const int rs = p->ref_sel[c];
            d.chain = c;
            for (int k = 0; k < 3; k++) { d.ref[k] = p->chains_host[c].rec[rs][k]; d.dec[k] = p->chains_host[c].rec[rs ^ 1][k]; }
            d.dep_progress = 0;
            p->ref_sel[c] ^= 1;
            if (p->host_rbsp[c] && p->host_mbrec[c])
            {
                /* one result per chain (slots_per_chain == 1): the finalizer exports it to host-mapped memory like a stream job, so
                 * the host reads NALs, flags and records without a device-to-host copy */
                d.arena_reset = 1;
                d.host_done = p->host_done + c;
                d.host_rbsp = p->host_rbsp[c]; d.host_rbsp_cap = p->host_rbsp_cap;
                d.host_mbrec = (h264e_mbrec_t *)p->host_mbrec[c];
                p->host_done[c].done = 0;
                p->slot_launch[c] = launch_id;
            }
        }
        d.chain_desc = p->chains_dev + d.chain;
        d.errflag = p->errflag;
        d.stepflags = p->stepflags + 2*c;
        d.frame_slot = t.frame_slot;
        d.first_row = (t.stream_mode && t.first_row > 0 && t.first_row < G.nmby) ? t.first_row : 0;
        d.narrow = t.stream_mode && t.narrow_window;
        any_narrow |= d.narrow;
        any_wide |= !d.narrow;
        d.hdr_nal = t.hdr_nal; d.hdr_nbits = t.hdr_nbits; d.hdr_bits = t.hdr_bits;
        {
            /* row bands exactly as the reference splits them (h264-lab.h:6530): mby += (nmby - mby)/(nthreads - ithr) */
            int mby = 0;
            d.nslices = t.nslices > 1 ? t.nslices : 1;
            for (int k = 0; k < d.nslices; k++) { d.slice_row[k] = (int16_t)mby; mby += (G.nmby - mby)/(d.nslices - k); }
            d.slice_row[d.nslices] = (int16_t)G.nmby;
        }
        d.clusters[0] = t.mv_clusters[0]; d.clusters[1] = t.mv_clusters[1];
        d.clusters_per_mb = 0;
        if (t.stream_mode && t.traj_from_device && p->traj_dev[t.slot])
            d.clusters_per_mb = p->traj_dev[t.slot] + (size_t)p->traj_cur[t.slot]*2*G.nmb;     /* the latest device walk of this slot */
        else if (t.mv_clusters_per_mb)
        {
            const size_t n = sizeof(int32_t)*2*(size_t)G.nmb;
            const int cs = t.stream_mode ? t.slot : c;
#ifdef H264E_EMU
            memcpy(p->clu_dev[cs], t.mv_clusters_per_mb, n);
#else
            /* the re-encode path is rare and synchronous: a blocking copy keeps the host array's lifetime simple */
            if (hipStreamSynchronize(p->stream) != hipSuccess || hipMemcpy(p->clu_dev[cs], t.mv_clusters_per_mb, n, hipMemcpyHostToDevice) != hipSuccess)
            {
                free(host);
                FAIL("mv_clusters upload failed");
            }
#endif
            d.clusters_per_mb = p->clu_dev[cs];
        }
        memcpy(d.qdat, t.qdat, sizeof(d.qdat));
        d.launch_id = launch_id;
        if (d.walk_on_device) p->traj_cur[t.slot] ^= 1;         /* this launch's walk writes the other buffer: it is the latest from now on */
    }
    if (!any) { free(host); return 0; }
    if (!p->group) device_acquire(p);   /* one launch at a time per device (see g_device_lock); a launch group owns the device as a whole (h264e_hip_group_join) */
    if (any_narrow && any_wide) { free(host); FAIL("submit: the jobs of one launch must agree on narrow_window"); }
    h264e_frame_task_t *slot = p->tasks_dev + (size_t)p->ring_pos*p->nchains;
    p->ring_pos = (p->ring_pos + 1) % TASK_RING;
    p->pending++;
#ifdef H264E_EMU
    memcpy(slot, host, sizeof(h264e_frame_task_t)*(size_t)p->nchains);
    free(host);
    for (int c = 0; c < p->nchains; c++)
    {
        const h264e_frame_task_t &T = slot[c];
        if (!T.active) continue;
        const ChainG C = chain_view(p->chains_dev[T.chain]);
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
                /* the three kernel variants, chosen like the launcher does */
                if (T.slice_type == 2 && all_intra) { row_prefetch<GEOM_INTRA>(*L, G, RT, row, x); row_step<GEOM_INTRA>(*L, G, C, RT, row, x, row0, row1); }
                else if (T.narrow) { row_prefetch<GEOM_NARROW>(*L, G, RT, row, x); row_step<GEOM_NARROW>(*L, G, C, RT, row, x, row0, row1); }
                else { row_prefetch<GEOM_WIDE>(*L, G, RT, row, x); row_step<GEOM_WIDE>(*L, G, C, RT, row, x, row0, row1); }
            }
            row_end(*L, G, C, row);
            free(L);
        }
        {
            int wstatus = 0, first_bad = -1;
            mv32 ws[2] = { T.exact_state[0], T.exact_state[1] };
            if (T.walk_on_device)
            {
                if (T.walk_prev)
                {
                    if (T.walk_prev->flag != T.launch_id || T.walk_prev->status != H264E_WALK_OK) wstatus = H264E_WALK_VOID;
                    else { ws[0] = T.walk_prev->state_out[0]; ws[1] = T.walk_prev->state_out[1]; }
                }
                if (!wstatus)
                {
                    first_bad = device_clusters_walk(G, T, C.mbrec + (size_t)T.frame_slot*G.nmb, ws, T.traj_out);
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
            finalize_frame(G, C, T, p->stepflags + 2*c);
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
#else
    HIPCHK(hipSetDevice(p->device));
    /* pageable source: the runtime stages the copy before returning, so `host` can be freed right away */
    hipError_t e = hipMemcpyAsync(slot, host, sizeof(h264e_frame_task_t)*(size_t)p->nchains, hipMemcpyHostToDevice, p->stream);
    if (e != hipSuccess) { free(host); FAIL("task upload: %s", hipGetErrorString(e)); }
    e = hipMemsetAsync(p->progress_all, 0, sizeof(int)*(size_t)p->nchains*G.nmby, p->stream);
    /* rows kept from the previous encode of a frame count as complete */
    for (int c = 0; c < p->nchains && e == hipSuccess; c++)
        if (host[c].active && host[c].first_row > 0)
        {
            int *done = (int *)malloc(sizeof(int)*(size_t)host[c].first_row);
            if (!done) { e = hipErrorOutOfMemory; break; }
            for (int r = 0; r < host[c].first_row; r++) done[r] = G.nmbx + 1;
            e = hipMemcpyAsync(p->chains_host[host[c].chain].progress, done, sizeof(int)*(size_t)host[c].first_row, hipMemcpyHostToDevice, p->stream);
            free(done);         /* pageable source: staged before the call returns */
        }
    if (e != hipSuccess) { free(host); FAIL("progress reset: %s", hipGetErrorString(e)); }
    /* wavefronts per macroblock row: two (search | reconstruction pipeline) halve the macroblock latency for twice the wave slots --
     * the better trade wherever a launch is latency bound (single-slice streams: mis-speculation events; rate control and the
     * frame-at-a-time API: a few frames per launch) and still level for multi-slice streams; an all-intra launch has nothing to
     * search and no events: one wave per row, twice the rows in flight (22.4 vs 18.3 M MB/s at 1080p) */
    /* (waves = 0 selects the intra-only variant of the one-wave kernel: no inter code, half the registers, twice the rows in flight) */
    /* (... and 4 the two-wave kernel allocated for 4 waves per SIMD: launches bound by the rows in flight -- 8K-class pictures, many slices) */
    /* measured with the final register allocation (gpurun_out/r3_lane4): 4 per SIMD wins wherever a launch offers enough rows to fill the
     * chip (8 slices 20.6 -> 23.3 M MB/s, 8K 6.6 -> 9.1 M, 4K 14.8 -> 15.3 M, 1080p single slice 9.56 -> 9.61 M); launches of a few frames
     * (rate control, the frame-at-a-time API) are pure latency and keep the 3-per-SIMD kernel with its fewer spills (10.5 vs 10.9 ms) */
    const int waves = p->waves ? p->waves : all_intra ? 0 : (njobs*G.nmby >= 1536) ? 4 : 2;
    (void)max_slices;
    if (p->group)
    {
        /* member of a launch group: the launch is merged with the other members' (group_launch_locked) */
        const int grc = group_submit(p, host, njobs, any_narrow, waves);
        free(host);
        return grc;
    }
    free(host);
    /* the dispatch order for this launch's shape (jobs up to the last active one; window geometry) */
    if (njobs != p->order_jobs || any_narrow != p->order_narrow)
    {
        if (build_order(p, njobs, any_narrow)) FAIL("out of host memory");
        HIPCHK(hipMemcpyAsync(p->order, p->order_host, sizeof(uint32_t)*(size_t)njobs*(G.nmby + 1), hipMemcpyHostToDevice, p->stream));     /* pageable: staged before the call returns */
        p->order_jobs = njobs; p->order_narrow = any_narrow;
    }
    const int pe = p->ev_pending;
    if (p->profile) HIPCHK(hipEventRecord(p->ev[pe][0], p->stream));
    launch_mb_kernel(G, any_narrow, waves, (unsigned)(njobs*(G.nmby + 1)), slot, p->order, p->stream);
    if (p->profile) HIPCHK(hipEventRecord(p->ev[pe][1], p->stream));
    HIPCHK(hipGetLastError());
    if (p->profile)
    {
        HIPCHK(hipEventRecord(p->ev[pe][2], p->stream));
        p->ev_pending++;
    }
#endif
    return 0;
}

extern "C" int h264e_hip_step_flags(h264e_hip_pool_t *p, int *flags /* [nchains][2] */)
{
    if (!p || !flags) FAIL("step_flags: bad argument");
#ifdef H264E_EMU
    memcpy(flags, p->stepflags, sizeof(int)*2*(size_t)p->nchains);
#else
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpy(flags, p->stepflags, sizeof(int)*2*(size_t)p->nchains, hipMemcpyDeviceToHost));
#endif
    return 0;
}

/* ---- streaming results: valid for pools created with slots_per_chain == 1 and tasks submitted with stream_mode */

extern "C" int h264e_hip_stream_done(h264e_hip_pool_t *p, int slot, h264e_hip_result_t *res)
{
    if (!p || slot < 0 || slot >= p->nchains || !p->host_rbsp[slot]) FAIL("stream_done: bad argument");
    const volatile h264e_hostdone_t *d = p->host_done + slot;
    const int v = d->done;
    if (v != p->slot_launch[slot] && v != -p->slot_launch[slot]) return 0;      /* not yet */
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (res)
    {
        res->walk_status = d->walk_status; res->first_bad = d->first_bad; res->state_out[0] = d->state_out[0]; res->state_out[1] = d->state_out[1];
        if (v > 0)
        {
            res->nbytes = d->nbytes; res->all_skipped = d->all_skipped; res->clusters_moved = d->clusters_moved; res->overflow = d->overflow; res->far_reads = d->far_reads;
            res->nslices = d->nslices; res->in_device = d->in_device;
            for (int k = 0; k < H264E_HIP_MAX_SLICES; k++) res->slice_nbytes[k] = d->slice_nbytes[k];
        }
    }
    return v > 0 ? 1 : 2;                                   /* 2: the job was aborted (or failed its own validation: walk_status) */
}

extern "C" int h264e_hip_stream_fetch_traj(h264e_hip_pool_t *p, int slot, int consumed, int32_t *dst)
{
    if (!p || !dst || slot < 0 || slot >= p->nchains || !p->traj_dev[slot]) FAIL("stream_fetch_traj: bad argument");
    const int32_t *src = p->traj_dev[slot] + (size_t)(p->traj_cur[slot] ^ (consumed ? 1 : 0))*2*p->G.nmb;
#ifdef H264E_EMU
    memcpy(dst, src, sizeof(int32_t)*2*(size_t)p->G.nmb);
#else
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpy(dst, src, sizeof(int32_t)*2*(size_t)p->G.nmb, hipMemcpyDeviceToHost));
#endif
    return 0;
}

extern "C" int h264e_hip_stream_copy_picture(h264e_hip_pool_t *p, int from, int to)
{
    if (!p || from < 0 || to < 0 || from >= p->nchains || to >= p->nchains) FAIL("stream_copy_picture: bad argument");
    if (from == to) return 0;
    const size_t plane = (size_t)p->G.W*p->G.H*3/2;
#ifdef H264E_EMU
    memcpy(p->chains_host[to].rec[0][0], p->chains_host[from].rec[0][0], plane);
#else
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpyAsync(p->chains_host[to].rec[0][0], p->chains_host[from].rec[0][0], plane, hipMemcpyDeviceToDevice, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
#endif
    return 0;
}

extern "C" const uint8_t *h264e_hip_stream_rbsp(h264e_hip_pool_t *p, int slot)
{
    return (p && slot >= 0 && slot < p->nchains) ? p->host_rbsp[slot] : 0;
}

/* a frame whose NALs did not fit the host mirror (res.in_device): copy them from the slot's device NAL arena; works while the
 * launch is still running (copy stream) */
extern "C" int h264e_hip_stream_fetch_nals(h264e_hip_pool_t *p, int slot, uint8_t *dst, uint32_t nbytes)
{
    if (!p || !dst || slot < 0 || slot >= p->nchains || !p->chains_host[slot].nal_arena || nbytes > p->chains_host[slot].nal_cap) FAIL("stream_fetch_nals: bad argument");
#ifdef H264E_EMU
    memcpy(dst, p->chains_host[slot].nal_arena, nbytes);
#else
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpyAsync(dst, p->chains_host[slot].nal_arena, nbytes, hipMemcpyDeviceToHost, p->copy_stream));
    HIPCHK(hipStreamSynchronize(p->copy_stream));
#endif
    return 0;
}

extern "C" int h264e_hip_download_i420(h264e_hip_pool_t *p, int first, int nframes, uint8_t *host)
{
    if (!p || !host || first < 0 || nframes < 0 || first + nframes > p->frames_resident) FAIL("download_i420: bad range");
#ifdef H264E_EMU
    memcpy(host, p->clip + p->frame_bytes*(size_t)first, p->frame_bytes*(size_t)nframes);
#else
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpy(host, p->clip + p->frame_bytes*(size_t)first, p->frame_bytes*(size_t)nframes, hipMemcpyDeviceToHost));
#endif
    return 0;
}

extern "C" const h264e_hip_mbrec_t *h264e_hip_stream_mbrec(h264e_hip_pool_t *p, int slot)
{
    return (p && slot >= 0 && slot < p->nchains) ? p->host_mbrec[slot] : 0;
}

/* ask every job of the most recent submit to stop (the launch drains quickly; sync afterwards) */
extern "C" int h264e_hip_stream_abort(h264e_hip_pool_t *p)
{
    if (!p || !p->abort_word) FAIL("stream_abort: bad argument");
    __atomic_store_n(p->abort_word, p->launch_counter, __ATOMIC_RELEASE);
#ifdef H264E_EMU
    *p->abort_dev = p->launch_counter;
#else
    /* the kernel polls a word in device memory: the launch id is written there BY VALUE on a stream of its own, next to the running
     * launch -- not behind the application's staging uploads on the copy stream (up to hundreds of MB), and not as a copy whose source
     * could have moved on to the next launch's id by the time it executes */
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)p->abort_dev, p->launch_counter, 1, p->abort_stream));
#endif
    return 0;
}

/* 1 while work submitted to the pool is still running */
extern "C" int h264e_hip_busy(h264e_hip_pool_t *p)
{
#ifdef H264E_EMU
    (void)p;
    return 0;
#else
    if (!p) return 0;
    (void)hipSetDevice(p->device);
    if (p->group && hipEventQuery(p->group->ev_done) == hipErrorNotReady) return 1;
    return hipStreamQuery(p->stream) == hipErrorNotReady;
#endif
}

extern "C" int h264e_hip_result(h264e_hip_pool_t *p, int chain, int slot, h264e_hip_result_t *res)
{
    if (!p || !res || chain < 0 || chain >= p->nchains || slot < 0 || slot >= p->slots) FAIL("result: bad argument");
    h264e_frameout_t f;
#ifdef H264E_EMU
    f = p->chains_host[chain].fout[slot];
#else
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpy(&f, p->chains_host[chain].fout + slot, sizeof(f), hipMemcpyDeviceToHost));
#endif
    res->nbytes = f.nbytes; res->all_skipped = f.all_skipped; res->clusters_moved = f.clusters_moved; res->overflow = f.overflow; res->far_reads = f.far_reads;
    res->nslices = f.nslices;
    for (int k = 0; k < H264E_HIP_MAX_SLICES; k++) res->slice_nbytes[k] = f.slice_nbytes[k];
    return 0;
}

extern "C" int h264e_hip_read_rbsp(h264e_hip_pool_t *p, int chain, int slot, uint8_t *dst, uint32_t cap)
{
    if (!p || !dst || chain < 0 || chain >= p->nchains || slot < 0 || slot >= p->slots) FAIL("read_rbsp: bad argument");
    h264e_frameout_t f;
#ifdef H264E_EMU
    f = p->chains_host[chain].fout[slot];
    if (f.nbytes > cap) FAIL("read_rbsp: destination too small");
    memcpy(dst, p->chains_host[chain].arena + f.offset, f.nbytes);
#else
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpy(&f, p->chains_host[chain].fout + slot, sizeof(f), hipMemcpyDeviceToHost));
    if (f.nbytes > cap) FAIL("read_rbsp: destination too small");
    HIPCHK(hipMemcpy(dst, p->chains_host[chain].arena + f.offset, f.nbytes, hipMemcpyDeviceToHost));
#endif
    return (int)f.nbytes;
}

extern "C" int h264e_hip_read_chain(h264e_hip_pool_t *p, int chain, int nslots, h264e_hip_result_t *res, uint32_t *offsets,
                                    uint8_t *arena_dst, uint32_t cap, uint32_t *used)
{
    if (!p || !res || !offsets || !arena_dst || chain < 0 || chain >= p->nchains || nslots < 0 || nslots > p->slots) FAIL("read_chain: bad argument");
    h264e_frameout_t *f = (h264e_frameout_t *)malloc(sizeof(h264e_frameout_t)*(size_t)(nslots ? nslots : 1));
    uint32_t cur = 0;
    if (!f) FAIL("out of host memory");
#ifdef H264E_EMU
    memcpy(f, p->chains_host[chain].fout, sizeof(h264e_frameout_t)*(size_t)nslots);
    memcpy(&cur, p->chains_host[chain].cursor, 4);
#else
    if (hipSetDevice(p->device) != hipSuccess ||
        hipMemcpy(f, p->chains_host[chain].fout, sizeof(h264e_frameout_t)*(size_t)nslots, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(&cur, p->chains_host[chain].cursor, 4, hipMemcpyDeviceToHost) != hipSuccess)
    {
        free(f);
        FAIL("read_chain: copy failed");
    }
#endif
    for (int i = 0; i < nslots; i++)
    {
        res[i].nbytes = f[i].nbytes; res[i].all_skipped = f[i].all_skipped; res[i].clusters_moved = f[i].clusters_moved; res[i].overflow = f[i].overflow;
        res[i].far_reads = f[i].far_reads; res[i].nslices = f[i].nslices;
        for (int k = 0; k < H264E_HIP_MAX_SLICES; k++) res[i].slice_nbytes[k] = f[i].slice_nbytes[k];
        offsets[i] = f[i].offset;
    }
    free(f);
    if (cur > cap) FAIL("read_chain: destination too small (%u > %u)", cur, cap);
#ifdef H264E_EMU
    memcpy(arena_dst, p->chains_host[chain].arena, cur);
#else
    HIPCHK(hipMemcpy(arena_dst, p->chains_host[chain].arena, cur, hipMemcpyDeviceToHost));
#endif
    if (used) *used = cur;
    return 0;
}

extern "C" int h264e_hip_read_mbrec(h264e_hip_pool_t *p, int chain, int slot, h264e_hip_mbrec_t *dst)
{
    if (!p || !dst || chain < 0 || chain >= p->nchains || slot < 0 || slot >= p->slots) FAIL("read_mbrec: bad argument");
    const size_t n = sizeof(h264e_mbrec_t)*(size_t)p->G.nmb;
#ifdef H264E_EMU
    memcpy(dst, p->chains_host[chain].mbrec + (size_t)slot*p->G.nmb, n);
#else
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpy(dst, p->chains_host[chain].mbrec + (size_t)slot*p->G.nmb, n, hipMemcpyDeviceToHost));
#endif
    return 0;
}

extern "C" int h264e_hip_read_mbrec_all(h264e_hip_pool_t *p, int chain, int nslots, h264e_hip_mbrec_t *dst)
{
    if (!p || !dst || chain < 0 || chain >= p->nchains || nslots < 0 || nslots > p->slots) FAIL("read_mbrec_all: bad argument");
    const size_t n = sizeof(h264e_mbrec_t)*(size_t)p->G.nmb*(size_t)nslots;
#ifdef H264E_EMU
    memcpy(dst, p->chains_host[chain].mbrec, n);
#else
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpy(dst, p->chains_host[chain].mbrec, n, hipMemcpyDeviceToHost));
#endif
    return 0;
}

extern "C" int h264e_hip_read_recon(h264e_hip_pool_t *p, int chain, uint8_t *dst)
{
    if (!p || !dst || chain < 0 || chain >= p->nchains) FAIL("read_recon: bad argument");
    const size_t n = (size_t)p->G.W*p->G.H*3/2;
    const uint8_t *src = p->chains_host[chain].rec[p->ref_sel[chain]][0];   /* after the swap: last reconstruction */
#ifdef H264E_EMU
    memcpy(dst, src, n);
#else
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpy(dst, src, n, hipMemcpyDeviceToHost));
#endif
    return 0;
}

extern "C" int h264e_hip_reset_results(h264e_hip_pool_t *p, int chain)
{
    if (!p || chain < 0 || chain >= p->nchains) FAIL("reset_results: bad argument");
#ifdef H264E_EMU
    memset(p->chains_host[chain].cursor, 0, 16);
#else
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemsetAsync(p->chains_host[chain].cursor, 0, 16, p->stream));
#endif
    return 0;
}

extern "C" int h264e_hip_rewind_frame(h264e_hip_pool_t *p, int chain, int slot)
{
    if (!p || chain < 0 || chain >= p->nchains || slot < 0 || slot >= p->slots) FAIL("rewind_frame: bad argument");
    p->ref_sel[chain] ^= 1;
    /* the frame's result is dropped too: the arena cursor goes back to where that result starts */
#ifdef H264E_EMU
    *p->chains_host[chain].cursor = p->chains_host[chain].fout[slot].offset;
#else
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpyAsync(p->chains_host[chain].cursor, &p->chains_host[chain].fout[slot].offset, sizeof(uint32_t), hipMemcpyDeviceToDevice, p->stream));
#endif
    return 0;
}

extern "C" int h264e_hip_selftest_nal_escape(h264e_hip_pool_t *p, const uint8_t *src, uint32_t n, uint8_t *dst, uint32_t cap, uint32_t *out_n)
{
    if (!p || !src || !dst || !out_n) FAIL("selftest_nal_escape: bad argument");
    const size_t sb = ((size_t)n + 64 + 15) & ~(size_t)15, db = ((size_t)cap + 15) & ~(size_t)15;
#ifdef H264E_EMU
    uint8_t *s = (uint8_t *)calloc(1, sb), *d = (uint8_t *)calloc(1, db + 16);
    int overflow = 0;
    if (!s || !d) { free(s); free(d); FAIL("out of host memory"); }
    memcpy(s, src, n);
    *out_n = nal_escape_copy(d, cap, s, n, overflow);
    if (!overflow) memcpy(dst, d, *out_n);
    free(s); free(d);
    return overflow ? 1 : 0;
#else
    uint8_t *buf = 0;
    uint32_t res[2] = { 0, 0 };
    HIPCHK(hipSetDevice(p->device));
    if (hipMalloc((void **)&buf, sb + db + 64) != hipSuccess) FAIL("selftest_nal_escape: device allocation failed");
    hipError_t e = hipMemset(buf, 0, sb + db + 64);
    if (e == hipSuccess) e = hipMemcpy(buf, src, n, hipMemcpyHostToDevice);
    if (e == hipSuccess)
    {
        hipLaunchKernelGGL(h264e_nal_escape_selftest_kernel, dim3(1), dim3(64), 0, p->stream, buf + sb, cap, (const uint8_t *)buf, n, (uint32_t *)(buf + sb + db));
        e = hipStreamSynchronize(p->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(res, buf + sb + db, sizeof(res), hipMemcpyDeviceToHost);
    if (e == hipSuccess && !res[1] && res[0] <= cap) e = hipMemcpy(dst, buf + sb, res[0], hipMemcpyDeviceToHost);
    (void)hipFree(buf);
    if (e != hipSuccess) FAIL("selftest_nal_escape: %s", hipGetErrorString(e));
    *out_n = res[0];
    return res[1] ? 1 : 0;
#endif
}

extern "C" int h264e_hip_selftest_stage(h264e_hip_pool_t *p, int stage, const uint8_t *in, uint32_t nin, const int *args /* [24] */, uint8_t *out, uint32_t nout)
{
    if (!p || !in || !args || !out || stage < 1 || stage > 8 || nin > STAGE_IN_MAX || nout > STAGE_OUT_MAX) FAIL("selftest_stage: bad argument");
#ifdef H264E_EMU
    uint8_t *bi = (uint8_t *)calloc(1, STAGE_IN_MAX + 64), *bo = (uint8_t *)calloc(1, STAGE_OUT_MAX + 64);
    StageLds *S = (StageLds *)calloc(1, sizeof(StageLds));
    RowLds *L = (RowLds *)calloc(1, sizeof(RowLds));
    if (!bi || !bo || !S || !L) { free(bi); free(bo); free(S); free(L); FAIL("out of host memory"); }
    memcpy(bi, in, nin);
    stage_selftest(*S, *L, stage, bi, args, bo);
    memcpy(out, bo, nout);
    free(bi); free(bo); free(S); free(L);
    return 0;
#else
    uint8_t *buf = 0;
    HIPCHK(hipSetDevice(p->device));
    if (hipMalloc((void **)&buf, STAGE_IN_MAX + STAGE_OUT_MAX + 256) != hipSuccess) FAIL("selftest_stage: device allocation failed");
    hipError_t e = hipMemset(buf, 0, STAGE_IN_MAX + STAGE_OUT_MAX + 256);
    if (e == hipSuccess) e = hipMemcpy(buf, in, nin, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(buf + STAGE_IN_MAX, args, STAGE_NARGS*sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess)
    {
        hipLaunchKernelGGL(h264e_stage_selftest_kernel, dim3(1), dim3(64), 0, p->stream, stage, (const uint8_t *)buf, (const int *)(buf + STAGE_IN_MAX), buf + STAGE_IN_MAX + 128);
        e = hipStreamSynchronize(p->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out, buf + STAGE_IN_MAX + 128, nout, hipMemcpyDeviceToHost);
    (void)hipFree(buf);
    if (e != hipSuccess) FAIL("selftest_stage: %s", hipGetErrorString(e));
    return 0;
#endif
}

/* diagnostic: per-phase cycle sums of the -DH264E_STAMPS build, summed over chains (zeros in the product build) */
extern "C" int h264e_hip_stamps_read(h264e_hip_pool_t *p, unsigned long long *dst /* [32] */, int reset)
{
    if (!p || !dst) FAIL("stamps_read: bad argument");
    memset(dst, 0, 32*sizeof(unsigned long long));
    for (int c = 0; c < p->nchains; c++)
    {
        unsigned long long t[32];
#ifdef H264E_EMU
        memcpy(t, p->chains_host[c].prof, sizeof(t));
        if (reset) memset(p->chains_host[c].prof, 0, sizeof(t));
#else
        HIPCHK(hipSetDevice(p->device));
        HIPCHK(hipMemcpy(t, p->chains_host[c].prof, sizeof(t), hipMemcpyDeviceToHost));
        if (reset) HIPCHK(hipMemset(p->chains_host[c].prof, 0, sizeof(t)));
#endif
        for (int i = 0; i < 32; i++) dst[i] += t[i];
    }
    return 0;
}

extern "C" void h264e_hip_profile(h264e_hip_pool_t *p, int enable)
{
    if (!p) return;
    p->profile = enable; p->prof_launches = 0; p->prof_mb_ms = p->prof_splice_ms = 0;
}

extern "C" int h264e_hip_profile_read(h264e_hip_pool_t *p, double *mb_ms, double *splice_ms, int *launches)
{
    if (!p) FAIL("profile_read: null pool");
    if (mb_ms) *mb_ms = p->prof_mb_ms;
    if (splice_ms) *splice_ms = p->prof_splice_ms;
    if (launches) *launches = p->prof_launches;
    return 0;
}

extern "C" int h264e_hip_timer_start(h264e_hip_pool_t *p)
{
    if (!p) FAIL("timer_start: null pool");
#ifndef H264E_EMU
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipEventRecord(p->ev_t0, p->stream));
#endif
    return 0;
}

extern "C" int h264e_hip_timer_stop(h264e_hip_pool_t *p, double *ms)
{
    if (!p || !ms) FAIL("timer_stop: null argument");
    *ms = 0;
#ifndef H264E_EMU
    float f = 0;
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipEventRecord(p->ev_t1, p->stream));
    HIPCHK(hipEventSynchronize(p->ev_t1));
    HIPCHK(hipEventElapsedTime(&f, p->ev_t0, p->ev_t1));
    *ms = f;
#endif
    return 0;
}
