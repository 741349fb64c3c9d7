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
 * HIP only (hipcc --offload-arch=gfx950).  The host side of the boundary is h264e_pool.h, included at the end; the test-only
 * emulation of tests/emu compiles the same kernel HEADERS with its own launch functions and never sees this file.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "enc_row.h"
#include "enc_selftest.h"
#include "../../include/h264e_hip.h"

#include <hip/hip_runtime.h>


/* ------------------------------------------------------------------ device code */



/* relaxed poll of one progress counter until it reaches `need`.  Returns 0 = reached, -1 = producer failed or the
 * bound expired, -2 = producer was aborted (negative counters are poison left behind by a row that stopped). */
#ifndef H264E_POLL_SLEEP
/* x 64 cycles between two polls of a counter in device memory.  Measured in round 4 (gpurun_out/sleep_ab, sleep2): 1, 2, 8, 16, 32 are level
 * within 0.3 % in every regime (stream, 8 slices, rate control, lone frame: the round trip of the poll itself is what a hand-off costs);
 * what changes is what the parked waves issue meanwhile: with 16 here and 4 for the LDS hand-off words below, the scalar instructions
 * per delivered macroblock go from 10.6 k to 9.2 k (LDS 1.86 k -> 1.66 k) at the same speed */
#define H264E_POLL_SLEEP 16
#endif
DEV int poll_progress(const GLOBAL_AS int *flag, int need, int &seen, unsigned spin_limit)
{
    unsigned spins = 0;
    for (;;)
    {
        seen = uni(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));     /* uni: the loop control is scalar */
        if (seen >= need) return 0;
        if (seen < 0) return seen;
        if (++spins > spin_limit) return -1;
        __builtin_amdgcn_s_sleep(H264E_POLL_SLEEP);
    }
}

/* a progress counter of this row, for other workgroups (relaxed, agent scope; the payload in front of it was stored write-through and drained).
 * Every lane stores the same value to the same word -- one memory request for the wave: a store under "lane 0 only" is a divergent
 * branch, and behind one the compiler keeps the surrounding wave-uniform state (loop counters, the bit writer) in vector registers. */
DEV void publish(GLOBAL_AS int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

/* ---- hand-off words of the two-wave pipeline (LDS): release / acquire at workgroup scope, polled with s_sleep */
DEV int flag_get(const int *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
DEV void flag_set(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
#ifndef H264E_LDS_SLEEP
#define H264E_LDS_SLEEP 4            /* x 64 cycles between two looks at a hand-off word in LDS (1, 2, 4 measured level, see H264E_POLL_SLEEP) */
#endif
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
        __builtin_amdgcn_s_sleep(H264E_LDS_SLEEP);
    }
}
/* what the search wave tells the reconstruction wave while it is still searching macroblock need - 1 (enc_mb.h inter_choose) */
struct SearchSignals
{
    RowLds *L;
    int need;
    bool help;          /* three waves per row: a helper wave searches the 8x8 partition type */
    DEVM void noskip() const { flag_set(&L->f_noskip, need); }
    DEVM void bound(int u) const { L->early_bound = u; flag_set(&L->f_bound, need); }
    DEVM bool helper() const { return help; }
    DEVM void t3_request(mv32 mv_best, int sad_best, const rect_t &lim) const
    {
        L->t3_mv_best = mv_best; L->t3_sad_best = sad_best;
        L->t3_lim[0] = lim.x0; L->t3_lim[1] = lim.y0; L->t3_lim[2] = lim.x1; L->t3_lim[3] = lim.y1;
        flag_set(&L->f_t3req, need);            /* release: the request and the predictor context copy (inter_choose) are visible with it */
    }
    DEVM bool t3_wait() const { return lds_wait(&L->f_t3done, need, &L->f_stop) == 0; }
};
/* the reconstruction wave's view of the inter decision of macroblock need - 1 (enc_row.h mb_intra_decide) */
struct InterFromSearchWave
{
    RowLds *L;
    int need;
    bool back_wave;     /* four waves per row: mb_recon_back of the previous macroblock runs on the fourth wave */
    /* ... which reads the deblocking inputs mb_decide is about to overwrite: wait until it is done with macroblock need - 2 */
    DEVM bool before_decide() const { return !back_wave || need < 2 || lds_wait(&L->f_wdone, need - 1, &L->f_stop) == 0; }
    DEVM bool ready() const { return uni(flag_get(&L->f_inter)) >= need; }
    DEVM bool wait_ready() const { return lds_wait(&L->f_inter, need, &L->f_stop) == 0; }
    DEVM int early_bound() const { return uni(flag_get(&L->f_bound)) >= need ? uni(L->early_bound) : 0x7fffffff; }
    DEVM bool wait_either(const int *f) const
    {
        for (unsigned spins = 0;; spins++)
        {
            if (uni(flag_get(&L->f_inter)) >= need || uni(flag_get(f)) >= need) return true;
            if (uni(flag_get(&L->f_stop))) return false;
            if (spins > LDS_SPIN_LIMIT) { flag_set(&L->f_stop, -1); return false; }
            __builtin_amdgcn_s_sleep(H264E_LDS_SLEEP);
        }
    }
    DEVM bool wait_noskip_or_ready() const { return wait_either(&L->f_noskip); }
    DEVM bool wait_bound_or_ready() const { return wait_either(&L->f_bound); }
};

/*
 * Grid: njobs x (nmby + 1) workgroups of one wavefront -- or of two (WAVES = 2): the macroblock loop as a two-stage pipeline, one
 * wavefront searching macroblock x + 1 while the other reconstructs and writes x (enc_row.h).  Workgroup `row < nmby` encodes macroblock row
 * `row` of its job's frame; workgroup `nmby` is the job's finalizer: it follows the frame's rows as they end -- splices each into the
 * slice (enc_row.h splice_frame), walks its records (exact mv_clusters validation, JobWalk) -- and, in streaming use, exports the result
 * to host-mapped memory and raises the job's done word, so the host consumes frames while later frames of the same launch are still
 * being encoded.  Padding entries of a banded dispatch order (H264E_ORDER_PAD) are workgroups that exit at once.
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
    const int wv = WAVES >= 2 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;     /* which wavefront of the workgroup */
#ifdef H264E_LDS_PAD
    /* diagnostic build (Makefile `halfres`): extra LDS per workgroup so that fewer workgroups fit a CU -- what single-stream throughput
     * does when the resident rows are halved with the macroblock latency unchanged (the price of a multi-wave workgroup) */
    __shared__ volatile char lds_pad[H264E_LDS_PAD];
    if (G.nmbx < 0) lds_pad[LANE] = 1;
#endif
    /* workgroups are dispatched in index order: `order` lists (job, row) by the step at which the row can start
     * (H264E_FRAME_LAG*job + 2*row), so the resident workgroups are the ones that can make progress */
    const uint32_t jr = order[blockIdx.x];
    if (jr == H264E_ORDER_PAD) return;        /* padding of a banded dispatch order (h264e_pool.h build_order) */
    const int job = (int)(jr >> 16), row = (int)(jr & 0xffffu);
    const h264e_frame_task_t &T = tasks[job];
    if (!T.active) return;
    const ChainG C = chain_view(*T.chain_desc);
    GLOBAL_AS int *errflag = (GLOBAL_AS int *)uniptr(T.errflag);

    if (row == G.nmby)
    {
        /* ---- finalizer (one wavefront; the second one of a two-wave workgroup has nothing to do here) */
        if (wv) return;
#ifdef H264E_STAMPS
        unsigned long long fz_t = wall_clock64();
#define FZ_STAMP(id) do { const unsigned long long n_ = wall_clock64(); if (LANE == 0 && C.prof) atomicAdd(C.prof + (id), n_ - fz_t); fz_t = n_; } while (0)
#else
#define FZ_STAMP(id) do { } while (0)
#endif
        int st = 0, seen = 0;
        GLOBAL_AS h264e_hostdone_t *hd = (GLOBAL_AS h264e_hostdone_t *)T.host_done;
        GLOBAL_AS h264e_walkrec_t *wout = (GLOBAL_AS h264e_walkrec_t *)T.walk_out;
        /* The finalizer follows its frame ROW BY ROW (enc_row.h JobWalk / splice_frame): a row whose counter says complete is spliced
         * while the rows below it are still being encoded, and walked (exact mv_clusters validation) as soon as the state in front of
         * the frame is known -- the verdict of the frame in front, which usually arrives when this frame is nearly done: the rows
         * completed by then are walked in one go (also while this wave waits for its next row: the frame-to-frame chain of verdicts
         * must carry the walk alone, never a wait or the splice), and the verdict goes out right behind the walk of the last row. */
        int wstatus = 0, first_bad = -1, have_prev = !(T.walk_on_device && T.walk_prev), published = 0;
        mv32 ws[2] = { T.exact_state[0], T.exact_state[1] };
        const GLOBAL_AS h264e_walkrec_t *wp = (const GLOBAL_AS h264e_walkrec_t *)T.walk_prev;
        const GLOBAL_AS h264e_mbrec_t *rec = C.mbrec + (size_t)T.frame_slot*G.nmb;
        JobWalk W;
        W.begin(ws);
        /* Row-band slices: every slice of every frame starts from the state in front of the STREAM (the parent's state never moves,
         * h264-lab.h:6526), so the walk needs nothing of the frame in front but its status, and only to publish: the chain of verdicts
         * carries no walk at all (it was what bounded an 8-slice stream: one walk of ~0.26 ms per frame, 3830 frames/s) */
        const bool state_fixed = T.nslices > 1;
        /* the verdict of the frame in front has arrived: take its state (or its failure) */
        const auto take_prev = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (uni(wp->status) != H264E_WALK_OK) wstatus = H264E_WALK_VOID;
            else if (!state_fixed) { ws[0] = (mv32)uni(wp->state_out[0]); ws[1] = (mv32)uni(wp->state_out[1]); W.begin(ws); }
            have_prev = 1;
        };
        /* this frame's verdict, for the frame behind */
        const auto publish_verdict = [&]() {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (LANE == 0 && wout)
            {
                wout->state_out[0] = ws[0]; wout->state_out[1] = ws[1]; wout->status = wstatus; wout->first_bad = first_bad;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&wout->flag, T.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            published = 1;
        };
        /* rows up to `upto` are complete: walk what the walk may have; behind the last row the verdict goes out.  Returns non-zero
         * when the frame does not stand (mismatch, or void because the frame in front does not) */
        const auto walk_upto = [&](int upto) -> int {
            if (!T.walk_on_device || published) return 0;
            if (!have_prev && uni(__hip_atomic_load(&wp->flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= T.launch_id) take_prev();
            if (!have_prev && !state_fixed) return 0;
            if (!wstatus)
            {
                W.rows(G, T, rec, (GLOBAL_AS mv32 *)T.traj_out, upto);
                if (W.next_row < G.nmby) return 0;
                if (!have_prev)
                {
                    /* walked to the end on the fixed state: the verdict still has to wait for the status of the frame in front */
                    int sf = 0;
                    if (poll_progress(&wp->flag, T.launch_id, sf, G.spin_limit)) { wstatus = H264E_WALK_VOID; have_prev = 1; }
                    else take_prev();
                }
            }
            if (!wstatus)
            {
                W.end(T);
                first_bad = W.first_bad;
                ws[0] = W.s[0]; ws[1] = W.s[1];
                wstatus = first_bad >= 0 ? H264E_WALK_BAD : H264E_WALK_OK;
            }
            publish_verdict();
            return wstatus != H264E_WALK_OK;
        };
        SpliceOut so;
        if (G.fz_wait_all) for (int r = G.nmby - 1; r >= 0 && !st; r--) st = poll_progress(C.progress + r, G.nmbx + 1, seen, G.spin_limit);
        if (!st) st = splice_frame(G, C, T, so, [&](int, int r) -> int {
            for (unsigned spins = 0;; spins++)
            {
                seen = uni(__hip_atomic_load(C.progress + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (seen >= G.nmbx + 1) break;
                if (seen < 0) return seen;
                if (spins > G.spin_limit) return -1;
                /* waiting for row r: the rows above it are complete (and acquired) -- is their walk due? */
                if (r > 0 && walk_upto(r - 1)) return 1;
                __builtin_amdgcn_s_sleep(H264E_POLL_SLEEP);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            return walk_upto(r);
        });
        FZ_STAMP(32);               /* followed the frame's rows: waits, the splice of every completed row, the walk once its start state was in */
        if (st < 0)
        {
            /* a row of the frame was stopped or failed */
            if (LANE == 0)
            {
                if (st == -1) *errflag = 1;
                if (T.walk_on_device && wout && !published)
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
        if (T.walk_on_device)
        {
            if (!published)
            {
                /* every row is spliced and the state in front of the frame is still out: wait for it, walk the frame in one go */
                if (!have_prev)
                {
                    int sf = 0;
                    if (poll_progress(&wp->flag, T.launch_id, sf, G.spin_limit)) { wstatus = H264E_WALK_VOID; have_prev = 1; }
                    else take_prev();
                }
                FZ_STAMP(33);       /* waited for the verdict of the frame in front */
                (void)walk_upto(G.nmby - 1);
                FZ_STAMP(34);       /* the exact walk behind the splice (the frame in front was late) */
            }
            if (wstatus != H264E_WALK_OK)
            {
                if (LANE == 0)
                {
                    /* a mismatch stops the whole launch through the abort word (a leaf's concerns nobody but itself) */
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
        finalize_commit(G, C, T, so, (GLOBAL_AS int *)T.stepflags);
        FZ_STAMP(35);               /* the last row's splice + the result record */
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
        FZ_STAMP(36);               /* NAL escaping + export to host-mapped memory */
#ifdef H264E_STAMPS
        if (LANE == 0 && C.prof) atomicAdd(C.prof + 37, 1ull);
#endif
        return;
    }

    /* ---- macroblock row */
    if (row < T.first_row) return;          /* kept from the previous encode of this frame (its counter already says complete) */
    if (job == 0 && row == G.test_stall_row) return;    /* fault injection: a producer that never publishes (tests/test_gpu_failures.py) */
    if (wv == 0) row_begin(L, G, C, T, row);
    if (WAVES >= 2) __syncthreads();         /* the one workgroup barrier: all wavefronts see the row's tables and initial state */
    const RowTask RT = rowtask_load(T);      /* the task's hot fields, once, in registers */
    int row0 = 0, row1 = G.nmby;            /* the slice (row band) this row belongs to */
    for (int k = 0; k < T.nslices; k++)
        if (row >= T.slice_row[k] && row < T.slice_row[k + 1]) { row0 = T.slice_row[k]; row1 = T.slice_row[k + 1]; }
    row0 = uni(row0); row1 = uni(row1);
    GLOBAL_AS int *my_progress = C.progress + row;
    /* bookkeeping: the wave that reconstructs adds the macroblocks this row got through, once, when the row ends or stops */
    GLOBAL_AS unsigned long long *mb_counter = (GLOBAL_AS unsigned long long *)uniptr(T.mb_counter);
    const auto count_mbs = [&](int n) { if (mb_counter && n > 0 && LANE == 0) __hip_atomic_fetch_add(mb_counter, (unsigned long long)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };

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
            /* the abort word lives in device memory (raised by a finalizer whose mv_clusters walk failed, or by the host): one L2 read per
             * macroblock, ISSUED here and looked at behind the window loads -- as is the first look at the row above's counter: three
             * round trips to L2 / HBM in flight together instead of one after the other (the records above are still loaded BEHIND the
             * counter that covers them, and behind the acquire) */
            const int abort_v = abort_word ? __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : launch_id - 1;
            const GLOBAL_AS int *above = C.progress + (WAVES >= 2 ? G.nmby : 0) + (row - 1);
            const int early = seen < need ? __hip_atomic_load(above, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : seen;
            /* temporal dependency first, then the loads that only need it (input, reference window) ... */
            if (seen_dep < need_dep)
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
            if (WAVES >= 2 && !st && x >= 2) st = lds_wait(&L.f_wdone, x - 1, &L.f_stop);
            STAMP(L, 23);
            if (!st) row_prefetch<GEOM>(L, G, RT, row, x);
            STAMP(L, 0);
            if (uni(abort_v) == launch_id) st = -2;
            /* ... so that their latency overlaps with the wait for the row above */
            /* two waves: the search needs only the VECTORS of the row above, which that row's reconstruction wave hands over as soon as
             * they are decided (its `decided` counter, behind the progress counters) -- a transform / CAVLC / deblocking earlier than
             * the rest of the record, which this row's reconstruction wave waits for */
            if (!st && seen < need)
            {
                seen = uni(early);
                if (seen < 0) st = seen;
                else if (seen < need) st = poll_progress(above, need, seen, G.spin_limit);
                if (!st) consumer_acquire();
            }
            STAMP(L, 13);
            if (!st) load_top<WAVES >= 2 ? LOAD_TOP_MV : LOAD_TOP_ALL>(L, L.mb[x & 1], G, C.bottom + (size_t)(row - 1)*G.nmbx, C.pend + (size_t)(row - 1)*G.nmbx, x, row > row0);
            /* two waves: the search of x starts from the predictor context the decision of x - 1 leaves behind */
            STAMP(L, 0);
            if (WAVES >= 2 && !st && x >= 1) st = lds_wait(&L.f_decided, x, &L.f_stop);
            STAMP(L, 6);
            if (st)
            {
                /* stop: leave poison in this row's counter so everything behind it stops too (-1 failure, -2 abort); with two waves the
                 * reconstruction wave is the one that publishes, so it also leaves the poison (after what it is publishing right now) */
                if (WAVES >= 2) { flag_set(&L.f_stop, st); return; }
                if (LANE == 0)
                {
                    if (st == -1) *errflag = 1;
                    __hip_atomic_store(my_progress, st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                count_mbs(x);
                return;
            }
            if (WAVES == 1) row_step<GEOM>(L, G, C, RT, row, x, row0, row1);
            else mb_search<GEOM>(L, L.mb[x & 1], G, RT, row, x, row0, SearchSignals{ &L, x + 1, WAVES >= 3 });
            if (WAVES >= 3 && uni(flag_get(&L.f_stop))) return;         /* stopped while waiting for the helper wave */
            {
                /* a far reference read of this macroblock waited in vain (rv_wait_rect): what it encoded is not trustworthy */
                const int ff = uni(L.far_fail[0]) | (WAVES == 1 ? uni(L.far_fail[1]) : 0);
                if (ff)
                {
                    if (WAVES >= 2) { flag_set(&L.f_stop, ff < -2 ? -2 : ff); return; }
                    if (LANE == 0)
                    {
                        if (ff == -1) *errflag = 1;
                        __hip_atomic_store(my_progress, ff < -2 ? -2 : ff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    count_mbs(x);
                    return;
                }
            }
            if (WAVES >= 2) { flag_set(&L.f_inter, x + 1); STAMP(L, 14); continue; }
            /* producer: every handed-off byte was stored write-through (sc1, wave.h cstore*): drain them, then the counter --
             * no agent-scope release (L2 write-back) needed */
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            publish(my_progress, x + 1);
            STAMP(L, 14);
        }
        if (WAVES >= 2)
        {
#ifdef H264E_STAMPS
            if (LANE < 32 && C.prof) atomicAdd(C.prof + LANE, L.prof[0][LANE]);
#endif
            return;
        }
    } else if (WAVES >= 3 && wv == 2)
    {
        /* ---- the helper wave of the three-wave variant (launches of one or a few frames, where only the latency counts and the chip
         * is empty): searches the 8x8 partition type of macroblock x when the search wave asks for it */
        for (int x = 0; x < G.nmbx; x++)
        {
            int req = 0, st = 0;
            for (unsigned spins = 0;; spins++)
            {
                /* f_t3req is monotonic and carries the macroblock index (+1) of the LATEST request: only an exact match is a request for x.
                 * A larger value means the search wave finished x without asking and has already asked for a later macroblock -- skip x
                 * (running the 8x8 search with the context of x and the request data of x + 1 would also write into the hand-off
                 * buffer the reconstruction wave is reading).  A request for x precedes f_inter of x, and the search wave does not get
                 * past x while its request is unanswered, so "f_inter >= x + 1 without a request" is final. */
                const int r3 = uni(flag_get(&L.f_t3req));
                if (r3 == x + 1) { req = 1; break; }
                if (r3 > x + 1) break;
                if (uni(flag_get(&L.f_inter)) >= x + 1) break;                      /* the search wave is done with x without asking */
                if (uni(flag_get(&L.f_stop))) { st = 1; break; }
                if (spins > LDS_SPIN_LIMIT) { flag_set(&L.f_stop, -1); st = 1; break; }
                __builtin_amdgcn_s_sleep(H264E_LDS_SLEEP);
            }
            if (st) return;
            if (!req) continue;
            MbCtx m;
            mb_ctx_init<GEOM>(m, L, G, RT, row, x, row0, 2);
            const rect_t lim = { uni(L.t3_lim[0]), uni(L.t3_lim[1]), uni(L.t3_lim[2]), uni(L.t3_lim[3]) };
            const mv32 mv_best = (mv32)uni(L.t3_mv_best);
            const int sad_best = uni(L.t3_sad_best);
            GRP_EACH(t)
            {
                if (t == 3) search_type(L, L.mb[x & 1], m, 3, mv_best, sad_best, lim, 0);
            }
            wave_sync();
            const int ff = uni(L.far_fail[2]);
            if (ff) { flag_set(&L.f_stop, ff < -2 ? -2 : ff); return; }
            flag_set(&L.f_t3done, x + 1);
        }
#ifdef H264E_STAMPS
        if (LANE < 32 && C.prof) atomicAdd(C.prof + LANE, L.prof[2][LANE]);
#endif
        return;
    } else if (WAVES == 4 && wv == 3)
    {
        /* ---- the fourth wave of the latency variant: deblocking and every store of macroblock x (mb_recon_back) while the
         * reconstruction wave is already at the intra candidates of x + 1; owns the row's progress counter */
        for (int x = 0; x < G.nmbx; x++)
        {
            const int st = lds_wait(&L.f_front, x + 1, &L.f_stop);
            if (st)
            {
                if (LANE == 0) __hip_atomic_store(my_progress, st == -2 ? -2 : -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        /* the row is being stopped: poison for everything behind it */
                return;
            }
            MbCtx m;
            mb_ctx_init<GEOM>(m, L, G, RT, row, x, row0, 1);
            m.type = uni(L.d_type[x & 1]);
            mb_recon_back<GEOM>(L, L.mb[x & 1], m, G, C, RT, row, x, row0, row1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            publish(my_progress, x + 1);
            flag_set(&L.f_wdone, x + 1);
        }
#ifdef H264E_STAMPS
        if (LANE < 32 && C.prof) atomicAdd(C.prof + LANE, L.prof[3][LANE]);
#endif
        return;
    } else
    {
        /* ---- the reconstruction wave of the two-wave pipeline: intra candidates + decision of x, then transform / CAVLC / deblocking /
         * stores of x while the search wave is already on x + 1 */
        GLOBAL_AS int *my_decided = my_progress + G.nmby;
        GLOBAL_AS h264e_mbbottom_t *rowrec = C.bottom + (size_t)row*G.nmbx;
        int seen = 0;
        for (int x = 0; x < G.nmbx; x++)
        {
            MbBuf &B = L.mb[x & 1];
            MbCtx m;
            mb_ctx_init<GEOM>(m, L, G, RT, row, x, row0, 1);
            int ff = 0;
            /* samples, contexts and pending lines of the row above: complete when its reconstruction wave has published x + 1 */
            const int need = row > row0 ? imin(x + 2, G.nmbx) : 0;
            if (seen < need)
            {
                ff = poll_progress(C.progress + (row - 1), need, seen, G.spin_limit);
                if (!ff) consumer_acquire();
                else flag_set(&L.f_stop, ff);
            }
            STAMP(L, 17);
            if (!ff)
            {
                load_top<LOAD_TOP_REST>(L, B, G, C.bottom + (size_t)(row - 1)*G.nmbx, C.pend + (size_t)(row - 1)*G.nmbx, x, row > row0);
                if (!mb_intra_decide(L, B, m, RT, InterFromSearchWave{ &L, x + 1, WAVES == 4 })) ff = uni(flag_get(&L.f_stop));
                else
                {
                    flag_set(&L.f_decided, x + 1);
                    /* the vectors of this macroblock's bottom row, for the search of the row below: on their way while the chroma
                     * prediction runs, published before the transform starts */
                    WAVE_FOR(l) { if (l < 4) cstore32((gu8 *)(rowrec + x) + 32 + 4*l, (uint32_t)B.mv_top[l]); }
                    const auto publish_decided = [&]() {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                        publish(my_decided, x + 1);
                    };
                    if (WAVES == 4)
                    {
                        /* deblocking and the macroblock's stores go to the fourth wave; this one moves on to the intra candidates of x + 1 */
                        mb_recon_front<GEOM>(L, B, m, G, C, RT, row, x, row0, row1, publish_decided);
                        L.d_type[x & 1] = m.type;
                    } else mb_recon_write<GEOM>(L, B, m, G, C, RT, row, x, row0, row1, publish_decided);
                    ff = uni(L.far_fail[1]);
                    if (ff) { ff = ff < -2 ? -2 : ff; flag_set(&L.f_stop, ff); }
                    else if (WAVES == 4) flag_set(&L.f_front, x + 1);
                }
            }
            if (ff)
            {
                if (LANE == 0)
                {
                    if (ff == -1) *errflag = 1;
                    __hip_atomic_store(my_decided, ff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (WAVES != 4) __hip_atomic_store(my_progress, ff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      /* (four waves: the fourth one owns the progress counter, poison included) */
                }
                count_mbs(x);
                return;
            }
            if (WAVES == 4) { STAMP(L, 14); continue; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            publish(my_progress, x + 1);
            flag_set(&L.f_wdone, x + 1);
            STAMP(L, 14);
        }
        /* four waves: the row ends when the fourth wave has stored the last macroblock */
        if (WAVES == 4 && lds_wait(&L.f_wdone, G.nmbx, &L.f_stop)) { count_mbs(G.nmbx); return; }
    }
    count_mbs(G.nmbx);
    row_end(L, G, C, row);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (LANE == 0) __hip_atomic_store(my_progress, G.nmbx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   /* row buffer + meta complete */
}


__global__ void __launch_bounds__(64) h264e_nal_escape_selftest_kernel(uint8_t *dst, uint32_t cap, const uint8_t *src, uint32_t n, uint32_t *out)
{
    int overflow = 0;
    const uint32_t w = nal_escape_copy((GLOBAL_AS uint8_t *)dst, cap, (const GLOBAL_AS uint8_t *)src, n, overflow);
    if (threadIdx.x == 0) { out[0] = w; out[1] = (uint32_t)overflow; }
}

__global__ void __launch_bounds__(64) h264e_stage_selftest_kernel(int stage, const uint8_t *in, const int *args, uint8_t *out)
{
    __shared__ StageLds S;
    __shared__ RowLds L;
    stage_selftest(S, L, stage, (const GLOBAL_AS uint8_t *)in, args, (GLOBAL_AS uint8_t *)out);
}

__global__ void h264e_synth_kernel(uint8_t *dst, int w, int h, int t, uint32_t seed)
{
    const int n = w*h*3/2;
    for (int i = (int)(blockIdx.x*blockDim.x + threadIdx.x); i < n; i += (int)(gridDim.x*blockDim.x))
        dst[i] = sv_sample(w, h, t, seed, i);
}

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

/* ------------------------------------------------------------------ launches (what h264e_pool.h calls) */

/* variant: 0 = intra frames only (one wave per row, 4 per SIMD), 1 = one wave per row, 2 = two waves per row (3 per SIMD), 3 = the latency variant: four
 * waves per row (search | reconstruction | 8x8 search helper | deblocking and stores), 4 = two waves per row at 4 per SIMD */
static void bk_launch_mb(const h264e_geom_t &G, int narrow, int variant, int njobs, unsigned nblocks, const h264e_frame_task_t *td, const uint32_t *od, hipStream_t st)
{
    const dim3 grid(nblocks);
    (void)njobs;
    if (variant == 0) hipLaunchKernelGGL((h264e_mb_kernel<GEOM_INTRA, 1, H264E_WPEI>), grid, dim3(64), 0, st, G, td, od);
    else if (variant == 4)
    {
        if (narrow) hipLaunchKernelGGL((h264e_mb_kernel<GEOM_NARROW, 2, 4>), grid, dim3(128), 0, st, G, td, od);
        else hipLaunchKernelGGL((h264e_mb_kernel<GEOM_WIDE, 2, 4>), grid, dim3(128), 0, st, G, td, od);
    } else if (variant == 3)
    {
        if (narrow) hipLaunchKernelGGL((h264e_mb_kernel<GEOM_NARROW, 4, 2>), grid, dim3(256), 0, st, G, td, od);
        else hipLaunchKernelGGL((h264e_mb_kernel<GEOM_WIDE, 4, 2>), grid, dim3(256), 0, st, G, td, od);
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

static void bk_launch_synth(uint8_t *dst, int w, int h, int t, uint32_t seed, hipStream_t st)
{
    hipLaunchKernelGGL(h264e_synth_kernel, dim3(1024), dim3(256), 0, st, dst, w, h, t, seed);
}
static void bk_launch_ssd(int n, const uint8_t *clip, size_t frame_bytes, int width, int height, int in0, int in_mod, const h264e_chain_dev_t *chains, int pic0, int pic_mod, int W,
                          unsigned long long *out, hipStream_t st)
{
    hipLaunchKernelGGL(h264e_ssd_kernel, dim3(64, (unsigned)n, 3), dim3(256), 0, st, clip, frame_bytes, width, height, in0, in_mod, chains, pic0, pic_mod, W, out);
}
static void bk_launch_nal_selftest(uint8_t *dst, uint32_t cap, const uint8_t *src, uint32_t n, uint32_t *out, hipStream_t st)
{
    hipLaunchKernelGGL(h264e_nal_escape_selftest_kernel, dim3(1), dim3(64), 0, st, dst, cap, src, n, out);
}
static void bk_launch_stage_selftest(int stage, const uint8_t *in, const int *args, uint8_t *out, hipStream_t st)
{
    hipLaunchKernelGGL(h264e_stage_selftest_kernel, dim3(1), dim3(64), 0, st, stage, in, args, out);
}

#include "h264e_pool.h"
