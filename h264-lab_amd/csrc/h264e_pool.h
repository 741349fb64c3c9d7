/*
 * h264e_pool.h -- host side of the device boundary (include/h264e_hip.h): pools, launch groups, submits, results.
 *
 * Written ONCE against the HIP runtime API and five launch functions (bk_launch_mb, bk_launch_synth, bk_launch_ssd,
 * bk_launch_nal_selftest, bk_launch_stage_selftest) that the including translation unit defines in front of it:
 *   - h264e_kernels.hip : the product -- the real HIP runtime, the kernels launched with hipLaunchKernelGGL;
 *   - tests/emu/emu_backend.cpp : the test-only emulation -- a host-memory stand-in for the handful of runtime calls used here
 *     (tests/emu/emu_hip.h) and launch functions that run the same kernel sources as lane loops, row after row.
 * Nothing in this file knows which of the two it is compiled into.
 */
#ifndef H264E_POOL_H
#define H264E_POOL_H

static thread_local char g_err[256];       /* per calling thread */
#define FAIL(...) do { snprintf(g_err, sizeof(g_err), __VA_ARGS__); return -1; } while (0)
extern "C" const char *h264e_hip_last_error(void) { return g_err; }

/* ------------------------------------------------------------------ host side: pool */

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) FAIL("%s: %s", #x, hipGetErrorString(e_)); } while (0)
static int dev_malloc(void **p, size_t n) { return hipMalloc(p, n ? n : 1) == hipSuccess ? 0 : -1; }
static void dev_free(void *p) { if (p) (void)hipFree(p); }

/* pinned, device-mapped, coherent host memory: the kernel writes results here while it runs, the host polls it */
static int host_malloc(void **p, size_t n)
{
    if (hipHostMalloc(p, n ? n : 1, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return -1;
    memset(*p, 0, n ? n : 1);
    return 0;
}
static void host_free(void *p) { if (p) (void)hipHostFree(p); }

#define TASK_RING 128
static int imin_h(int a, int b) { return a < b ? a : b; }

/* One launch at a time per device, process-wide.  The macroblock kernel's forward-progress argument (every workgroup waits for
 * workgroups dispatched before it, which are resident or finished) assumes the launch has the device's wave slots to itself: two
 * such launches side by side can fill the slots with waiting workgroups of one while the workgroups they wait for sit undispatched
 * behind the other's (measured: "bounded spin expired" with 3-4 concurrent clip encoders, tools/multi_clip_probe.py).  A pool takes
 * its device's token with its first submit and gives it back when its launches have drained (h264e_hip_sync / release / destroy). */
#include <pthread.h>
#define H264E_MAX_DEVICES 64
/* a token, not a mutex: a launch group takes it on the thread that launches the merged grid and gives it back on whichever member
 * thread sees the launch drained -- a pthread mutex may only be unlocked by the thread that locked it */
static pthread_mutex_t g_device_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_device_cv = PTHREAD_COND_INITIALIZER;
static int g_device_held[H264E_MAX_DEVICES];
static void device_token_take(int device)
{
    const unsigned d = (unsigned)device % H264E_MAX_DEVICES;
    pthread_mutex_lock(&g_device_mu);
    while (g_device_held[d]) pthread_cond_wait(&g_device_cv, &g_device_mu);
    g_device_held[d] = 1;
    pthread_mutex_unlock(&g_device_mu);
}
static void device_token_give(int device)
{
    const unsigned d = (unsigned)device % H264E_MAX_DEVICES;
    pthread_mutex_lock(&g_device_mu);
    g_device_held[d] = 0;
    pthread_cond_broadcast(&g_device_cv);
    pthread_mutex_unlock(&g_device_mu);
}

/*
 * One encoder PROCESS per device.  The launch lock above only orders the launches of one process; a second process on the same GPU
 * would put its persistent launches next to ours (bounded spins expire, launches are repeated: slow, never wrong).  The first pool a
 * process creates on a device therefore takes an advisory lock on a file named after the device's PCI bus id and keeps it until its
 * last pool on that device is gone; a second process fails fast with a message that says who holds the device.
 * H264E_SHARE_DEVICE=1 skips the guard (e.g. to run two small encoders side by side on purpose).
 */
#include <errno.h>
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
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
        /* O_CLOEXEC: the lock must not leak into fork+exec children (the device would stay "in use" while an unrelated child lives);
         * O_NOFOLLOW: a planted symlink is refused, not followed and truncated; the file is made world-usable so that another user's
         * encoder can take the same lock -- and when it still cannot be opened for that reason, somebody else's encoder owns it */
        int fd = open(path, O_RDWR | O_CREAT | O_CLOEXEC | O_NOFOLLOW, 0666);
        if (fd >= 0) (void)fchmod(fd, 0666);
        if (fd < 0 && (errno == EACCES || errno == EPERM)) fd = open(path, O_RDONLY | O_CLOEXEC | O_NOFOLLOW);      /* flock needs no write access */
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
                if (ftruncate(fd, 0) == 0 && write(fd, me, (size_t)n) != n) { /* the pid is informational (and not writable through a read-only descriptor) */ }
                g_guard_fd[device] = fd;
            }
        } else if (errno == EACCES || errno == EPERM)
        {
            snprintf(g_err, sizeof(g_err), "device %d: lock file %.100s belongs to another user (their encoder owns the device); H264E_SHARE_DEVICE=1 overrides", device, path);
            g_guard_pools[device]--;
            rc = -1;
        } else if (errno != ENOENT && errno != ENOTDIR)
        {
            snprintf(g_err, sizeof(g_err), "device %d: cannot open lock file %.100s: %.40s (see H264E_LOCK_DIR, H264E_SHARE_DEVICE)", device, path, strerror(errno));
            g_guard_pools[device]--;
            rc = -1;
        }       /* (no lock directory at all: no guard) */
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
    unsigned long long *mb_counter;      /* device: macroblocks reconstructed by this pool's rows, delivered or not (h264e_hip_mb_counter) */
    uint32_t *order;                     /* device [nchains*(nmby+1)] (job << 16) | row in dispatch order of the current launch shape */
    uint32_t *order_host;                /* host copy being built (build_order) */
    int order_jobs, order_narrow, order_sliced;   /* the launch shape `order` holds: jobs, window geometry, row-band slices or not (-1: none yet) */
    size_t order_count;                  /* ... and its entries = the launch's workgroups (banded orders carry padding) */
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
    hipStream_t stream;
    hipStream_t copy_stream;             /* uploads that overlap with kernels on `stream` */
    hipStream_t abort_stream;            /* carries nothing but abort requests (h264e_hip_stream_abort) */
    hipEvent_t ev_t0, ev_t1, ev_prep;
    hipEvent_t ev[TASK_RING][3];         /* per pending submit: before / between / after the two kernels */
    int ev_pending;
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
    pthread_mutex_t mu;
    pthread_cond_t cv;
    hipStream_t stream;
    hipEvent_t ev_done, ev_t0[2], ev_t1[2];     /* launch times per window geometry (a round has at most one launch of each) */
    int timed[2];                        /* which of the two launched in the last round */
    h264e_frame_task_t *tasks_dev; size_t tasks_cap;
    uint32_t *order_dev; size_t order_cap;
    int holds_device;                    /* the group's merged launch owns the device's launch token (taken at the launch, given back when it has drained) */
    int last_variant[2];                 /* kernel variant of the last round's launches, per window geometry (-1: none): for tests and logs */
    char err[256];                       /* why the last round failed: every member reports it, not only the thread that launched */
};

extern "C" int h264e_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static void device_acquire(h264e_hip_pool_t *p)
{
    if (p->holds_device) return;
    device_token_take(p->device);
    p->holds_device = 1;
}
static void device_release(h264e_hip_pool_t *p)
{
    if (!p->holds_device) return;
    p->holds_device = 0;
    device_token_give(p->device);
}

extern "C" void h264e_hip_pool_destroy(h264e_hip_pool_t *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    if (p->copy_stream) (void)hipStreamSynchronize(p->copy_stream);
    if (p->abort_stream) (void)hipStreamSynchronize(p->abort_stream);
    device_release(p);
    host_free(p->hheap);
    free(p->host_rbsp); free(p->host_mbrec); free(p->slot_launch); free(p->order_host);
    dev_free(p->heap);
    if (p->stream)
    {
        for (int i = 0; i < TASK_RING; i++) for (int k = 0; k < 3; k++) (void)hipEventDestroy(p->ev[i][k]);
        (void)hipEventDestroy(p->ev_t0); (void)hipEventDestroy(p->ev_t1); (void)hipEventDestroy(p->ev_prep);
        (void)hipStreamDestroy(p->stream);
        if (p->copy_stream) (void)hipStreamDestroy(p->copy_stream);
        if (p->abort_stream) (void)hipStreamDestroy(p->abort_stream);
    }
    free(p->chains_host); free(p->clu_dev); free(p->ref_sel); free(p->traj_dev); free(p->traj_cur);
    if (p->guarded) process_guard_release(p->device);
    free(p);
}

/* Dispatch order of a launch of `jobs` jobs: (job, row) sorted by the step at which the row can start when consecutive jobs are
 * consecutive frames of one stream (lag*job + 2*row: a counting sort); every workgroup still only waits for workgroups that precede it
 * in this order (far reads: a bounded distance ahead).  Built for the number of jobs a launch really has, so that a pool with many
 * slots does not dispatch thousands of empty workgroups with every short launch.
 * H264E_XCD_BANDS=N (experiment, profiles/r02_xcd_bands.txt): workgroups are dealt round-robin over the 8 XCDs (MI355X_MICROARCH.md:
 * blocks b and b+8 share one), so the order is additionally arranged so that a macroblock row lands on the XCD of its band of rows
 * (row*N/nmby mod 8): the rows whose reference windows overlap then share an L2. */
/* entries a launch of `jobs` jobs can take in the dispatch order (banded: eight equally long queues per job, padded) */
static size_t order_capacity(const h264e_geom_t &G, int jobs) { return (size_t)jobs*(size_t)(8*((G.nmby + 7)/8 + 1)); }
static int build_order(h264e_hip_pool_t *p, int jobs, int narrow, int sliced)
{
    const h264e_geom_t &G = p->G;
    const int rows = G.nmby + 1, total = jobs*rows, lag = narrow ? H264E_NARROW_FRAME_LAG : H264E_FRAME_LAG, maxkey = lag*(jobs - 1) + 2*(rows - 1);
    /* measured (gpurun_out/bands_pad, one MI355X, padded queues): 8 bands nearly halve FETCH_SIZE everywhere (1080p: 1521 -> 830 MB per
     * launch, WRITE_SIZE 765 -> 603) -- HBM traffic nobody waits for at 2 % of the bandwidth -- and what they do to the SPEED depends on
     * what a launch is short of: single-slice streams of big pictures gain (4K +5 %, 8K +14 %: a frame's rows no longer fit the L2s at
     * random), everything else loses 1-13 % (1080p -4 %, 720p -1 %, CIF -13 %, rate control -4 %; row-band slices -9 % at every size,
     * 4K and 8K included: a slice is a band, and an XCD cannot share its slice's load with the others).  So: on for single-slice
     * launches from 4K up, H264E_XCD_BANDS=8 / 0 forces */
    const int bands = getenv("H264E_XCD_BANDS") ? atoi(getenv("H264E_XCD_BANDS")) : (G.nmb >= 30000 && !sliced ? 8 : 0);
    uint32_t *ord = p->order_host;
    int *start = (int *)calloc((size_t)maxkey + 2, sizeof(int));
    uint32_t *tmp = bands ? (uint32_t *)malloc(sizeof(uint32_t)*(size_t)total) : ord;
    if (!start || !tmp) { free(start); if (bands) free(tmp); return -1; }
    for (int job = 0; job < jobs; job++) for (int r = 0; r < rows; r++) start[lag*job + 2*r + 1]++;
    for (int k = 0; k <= maxkey; k++) start[k + 1] += start[k];
    for (int job = 0; job < jobs; job++) for (int r = 0; r < rows; r++) tmp[start[lag*job + 2*r]++] = ((uint32_t)job << 16) | (uint32_t)r;     /* ties: by job */
    free(start);
    p->order_count = (size_t)total;
    if (bands)
    {
        /* Eight queues in key order, one per XCD; slot i takes entry i / 8 of queue i % 8.  The queues are EQUALLY LONG, job by job:
         * a band that has fewer rows than the tallest one (nmby is rarely a multiple of 8; one band also carries the finalizer) is
         * padded with entries that are nobody's.  Without that the queues drift apart by a row or two per job, and because workgroups
         * are dispatched strictly in index order, an XCD whose resident workgroups all wait for rows of a queue that lags behind
         * blocks the dispatch of exactly those rows: measured as 2-4 % at 1080p with fixed bands, and as a dead launch ("bounded spin
         * expired") once a launch is long enough -- 600 jobs of 720p, CIF, or 1080p with 8 slices. */
        const int per = (G.nmby + 7)/8 + 1;
        const size_t qlen = (size_t)per*jobs;
        uint32_t *q = (uint32_t *)malloc(sizeof(uint32_t)*8*qlen);
        int *fill = (int *)calloc((size_t)8*jobs, sizeof(int));
        size_t cnt[8] = { 0 };
        if (!q || !fill) { free(q); free(fill); free(tmp); return -1; }
        for (size_t i = 0; i < 8*qlen; i++) q[i] = H264E_ORDER_PAD;
        /* queue x, job j owns the entries [j*per, (j+1)*per) ... in KEY order that would interleave the jobs; so: append in key order, and
         * when a job's last entry of a queue has gone in, append its padding right behind it */
        int *want = (int *)calloc((size_t)8*jobs, sizeof(int));
        if (!want) { free(q); free(fill); free(tmp); return -1; }
        for (int i = 0; i < total; i++)
        {
            const int row = (int)(tmp[i] & 0xffffu), x = (row >= G.nmby ? bands - 1 : imin_h(bands - 1, row*bands/G.nmby)) & 7;      /* band b -> XCD b % 8 */
            want[8*(tmp[i] >> 16) + x]++;
        }
        int empty = 0;
        for (int i = 0; i < 8*jobs; i++) if (!want[i]) empty = 1;
        if (empty)
        {
            /* fewer bands than XCDs (tiny pictures, H264E_XCD_BANDS < 8): no banding */
            memcpy(ord, tmp, sizeof(uint32_t)*(size_t)total);
            free(q); free(fill); free(want); free(tmp);
            return 0;
        }
        for (int i = 0; i < total; i++)
        {
            const int row = (int)(tmp[i] & 0xffffu), jb = (int)(tmp[i] >> 16), x = (row >= G.nmby ? bands - 1 : imin_h(bands - 1, row*bands/G.nmby)) & 7;
            q[(size_t)x*qlen + cnt[x]++] = tmp[i];
            if (++fill[8*jb + x] == want[8*jb + x]) cnt[x] += (size_t)(per - want[8*jb + x]);        /* the padding stays H264E_ORDER_PAD */
        }
        size_t n = 0;
        for (size_t k = 0; k < qlen; k++) for (int x = 0; x < 8; x++) ord[n++] = q[(size_t)x*qlen + k];
        p->order_count = n;
        free(q); free(fill); free(want); free(tmp);
    }
    return 0;
}

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
    G.fz_wait_all = getenv("H264E_FZ_WAIT_ALL") ? atoi(getenv("H264E_FZ_WAIT_ALL")) : 0;
    p->test_upload_fail_at = -1;
    if (knobs)
    {
        if (getenv("H264E_TEST_ROW_BYTES_PER_MB")) { const int b = atoi(getenv("H264E_TEST_ROW_BYTES_PER_MB"))/4; G.row_words = G.nmbx*(b > 1 ? b : 1); }
        if (getenv("H264E_TEST_SPIN_LIMIT")) G.spin_limit = (unsigned)atol(getenv("H264E_TEST_SPIN_LIMIT"));
        if (getenv("H264E_TEST_STALL_ROW")) G.test_stall_row = atoi(getenv("H264E_TEST_STALL_ROW"));
        if (getenv("H264E_TEST_UPLOAD_FAIL_AT")) p->test_upload_fail_at = atoi(getenv("H264E_TEST_UPLOAD_FAIL_AT"));
    }
    p->frame_bytes = (size_t)width*height*3/2;
    p->waves = getenv("H264E_WAVES") ? atoi(getenv("H264E_WAVES")) : 0;                  /* 1 / 2 / 3 / 4: forced (A-B measurements, tests); else chosen per launch */
    if (p->waves != 1 && p->waves != 2 && p->waves != 3 && p->waves != 4) p->waves = 0;
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
        p->progress_all = (int *)carve(sizeof(int)*2*(size_t)nchains*G.nmby, 256);     /* per slot: nmby row counters, then nmby `decided` counters */
        p->errflag = (int *)carve(sizeof(int), 256);
        p->mb_counter = (unsigned long long *)carve(sizeof(unsigned long long), 256);
        p->stepflags = (int *)carve(sizeof(int)*2*(size_t)nchains, 256);
        p->abort_dev = (int *)carve(64, 256);
        p->walkrec = (h264e_walkrec_t *)carve(sizeof(h264e_walkrec_t)*(size_t)nchains, 256);
        p->ssd_dev = (unsigned long long *)carve(sizeof(unsigned long long)*3*(size_t)nchains, 256);
        p->order = (uint32_t *)carve(sizeof(uint32_t)*order_capacity(G, nchains), 256);
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
            C.progress = p->progress_all + (size_t)c*2*G.nmby;
            C.rowbits = (uint32_t *)carve(sizeof(uint32_t)*(size_t)G.nmby*G.row_words, 256);
            C.rowmeta = (h264e_rowmeta_t *)carve(sizeof(h264e_rowmeta_t)*(size_t)G.nmby, 256);
            C.mbrec = (h264e_mbrec_t *)carve(sizeof(h264e_mbrec_t)*(size_t)G.nmb*slots, 256);
            C.arena = (uint8_t *)carve(arena_cap, 256);
            C.arena_cap = arena_cap;
            C.nal_arena = slots == 1 ? (uint8_t *)carve(nal_cap, 256) : 0;
            C.nal_cap = slots == 1 ? nal_cap : 0;
            C.cursor = (uint32_t *)carve(16, 256);
            C.fout = (h264e_frameout_t *)carve(sizeof(h264e_frameout_t)*(size_t)slots, 256);
            C.prof = (unsigned long long *)carve(sizeof(unsigned long long)*48, 256);     /* 0..31 the rows' phases, 32..47 the finalizer's */
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
    (void)hipMemset(p->heap, 0, p->heap_bytes);
    if (hipMemcpy(p->chains_dev, p->chains_host, sizeof(h264e_chain_dev_t)*(size_t)nchains, hipMemcpyHostToDevice) != hipSuccess)
    {
        h264e_hip_pool_destroy(p);
        FAIL("descriptor upload failed");
    }
    p->order_host = (uint32_t *)malloc(sizeof(uint32_t)*order_capacity(G, nchains));
    if (!p->order_host) { h264e_hip_pool_destroy(p); FAIL("out of host memory"); }
    p->order_jobs = -1;
    *pool = p;
    return 0;
}

extern "C" int h264e_hip_upload_i420(h264e_hip_pool_t *p, int first, int nframes, const uint8_t *host)
{
    if (!p || first < 0 || nframes < 0 || first + nframes > p->frames_resident) FAIL("upload_i420: bad range");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpyAsync(p->clip + p->frame_bytes*(size_t)first, host, p->frame_bytes*(size_t)nframes, hipMemcpyHostToDevice, p->stream));
    return 0;
}

extern "C" int h264e_hip_upload_i420_async(h264e_hip_pool_t *p, int first, int nframes, const uint8_t *host)
{
    if (!p || first < 0 || nframes < 0 || first + nframes > p->frames_resident) FAIL("upload_i420_async: bad range");
    if (p->async_uploads++ == p->test_upload_fail_at) FAIL("upload_i420_async: injected failure (H264E_TEST_UPLOAD_FAIL_AT)");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpyAsync(p->clip + p->frame_bytes*(size_t)first, host, p->frame_bytes*(size_t)nframes, hipMemcpyHostToDevice, p->copy_stream));
    return 0;
}

extern "C" int h264e_hip_upload_wait(h264e_hip_pool_t *p)
{
    if (!p) FAIL("upload_wait: null pool");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->copy_stream));
    return 0;
}

extern "C" int h264e_hip_upload_busy(h264e_hip_pool_t *p)
{
    if (!p) return 0;
    (void)hipSetDevice(p->device);
    const hipError_t e = hipStreamQuery(p->copy_stream);
    if (e == hipErrorNotReady) return 1;
    if (e != hipSuccess) FAIL("upload: %s", hipGetErrorString(e));      /* -1: the copy was lost, not finished */
    return 0;
}

extern "C" void *h264e_hip_host_alloc(size_t bytes)
{
    void *q = 0;
    return hipHostMalloc(&q, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? q : 0;
}

extern "C" void h264e_hip_host_free(void *q)
{
    if (q) (void)hipHostFree(q);
}


extern "C" int h264e_hip_ssd_frames(h264e_hip_pool_t *p, int n, int in0, int in_mod, int pic0, int pic_mod, uint64_t *out)
{
    if (!p || !out || n <= 0 || n > p->nchains || in_mod <= 0 || in_mod > p->frames_resident || pic_mod <= 0 || pic_mod > p->nchains) FAIL("ssd_frames: bad argument");
    const h264e_geom_t &G = p->G;
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemsetAsync(p->ssd_dev, 0, sizeof(unsigned long long)*3*(size_t)n, p->stream));
    bk_launch_ssd(n, (const uint8_t *)p->clip, p->frame_bytes, G.width, G.height, in0, in_mod, (const h264e_chain_dev_t *)p->chains_dev, pic0, pic_mod, G.W, p->ssd_dev, p->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, p->ssd_dev, sizeof(unsigned long long)*3*(size_t)n, hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    return 0;
}

extern "C" int h264e_hip_read_recon_slot(h264e_hip_pool_t *p, int slot, uint8_t *dst)
{
    if (!p || !dst || slot < 0 || slot >= p->nchains) FAIL("read_recon_slot: bad argument");
    const size_t n = (size_t)p->G.W*p->G.H*3/2;
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpy(dst, p->chains_host[slot].rec[0][0], n, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int h264e_hip_upload_planes(h264e_hip_pool_t *p, int index, const uint8_t *const yuv[3], const int stride[3])
{
    if (!p || index < 0 || index >= p->frames_resident) FAIL("upload_planes: bad index");
    uint8_t *d = p->clip + p->frame_bytes*(size_t)index;
    for (int c = 0; c < 3; c++)
    {
        const int w = p->G.width >> (c ? 1 : 0), h = p->G.height >> (c ? 1 : 0);
        HIPCHK(hipSetDevice(p->device));
        HIPCHK(hipMemcpy2DAsync(d, (size_t)w, yuv[c], (size_t)stride[c], (size_t)w, (size_t)h, hipMemcpyHostToDevice, p->stream));
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
        HIPCHK(hipSetDevice(p->device));
        bk_launch_synth(d, p->G.width, p->G.height, t0 + i, seed, p->stream);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int h264e_hip_sync(h264e_hip_pool_t *p)
{
    if (!p) FAIL("sync: null pool");
    HIPCHK(hipSetDevice(p->device));
    if (p->group)
    {
        /* the merged launch of the round this pool submitted in (all members' jobs) */
        h264e_hip_group_t *g = p->group;
        const hipError_t eg = hipEventSynchronize(g->ev_done);
        /* the merged launch has drained (or is lost): the first member to see that gives the device's launch token back */
        pthread_mutex_lock(&g->mu);
        if (g->holds_device && !g->arrived) { g->holds_device = 0; device_token_give(g->device); }
        pthread_mutex_unlock(&g->mu);
        if (eg != hipSuccess) FAIL("group launch: %s", hipGetErrorString(eg));
        if (p->profile && g->nmembers && g->member[0] == p)
            for (int k = 0; k < 2; k++)
            {
                float a = 0;
                if (g->timed[k] && hipEventElapsedTime(&a, g->ev_t0[k], g->ev_t1[k]) == hipSuccess) { p->prof_mb_ms += a; p->prof_launches++; }
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
    device_release(p);
    p->pending = 0;
    return 0;
}

/* give the device back after a failure in the middle of a launch sequence (no error reporting of its own) */
extern "C" void h264e_hip_release(h264e_hip_pool_t *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    device_release(p);
}


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
    pthread_mutex_init(&g->mu, 0); pthread_cond_init(&g->cv, 0);
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&g->stream) != hipSuccess) { free(g); FAIL("group_create: no stream on device %d", device); }
    (void)hipEventCreate(&g->ev_done);
    for (int k = 0; k < 2; k++) { (void)hipEventCreate(&g->ev_t0[k]); (void)hipEventCreate(&g->ev_t1[k]); g->last_variant[k] = -1; }
    *out = g;
    return 0;
}

/* Kernel variant of a launch that offers `rows` macroblock rows in `wgs` workgroups (bk_launch_mb): 0 intra frames only; 4 two waves per
 * row at 4 per SIMD (2048 resident workgroups) where the rows fill the chip; 3 the four-wave latency variant (192 VGPRs: 512 resident
 * workgroups) ONLY where the whole grid is resident anyway -- its far reads may wait for any workgroup of the grid; 2 in between. */
#define H264E_RESIDENT_WG_V2 1536
#define H264E_RESIDENT_WG_V3 512
/* `parallel`: the launch has more independent work than one single-slice stream's temporal wavefront -- several slices per frame,
 * pictures of 200+ macroblock rows (8K class), several streams merged: only there do the extra resident rows of the 4-per-SIMD kernel
 * buy more than its 128-register allocation costs (spills in the search; r4 sweep gpurun_out/var_sweep*: 1080p 2 / 4 / 8 slices +5 %,
 * 8K single slice +34 %, two slices +20 %; a single-slice 1080p / 4K / 720p stream -6 % / -5 % / -3 % -- its passes are bound by the
 * mis-speculation refills, i.e. by the macroblock latency).  `tree`: the launch carries hedge leaves (rate control): a few frames of
 * it are ever used, pure latency (1080p 4 Mbit/s: 245 vs 213 fps, 8 slices 413 vs 381). */
static int pick_variant(int forced, int all_intra, int rows, int wgs, int parallel, int tree)
{
    if (forced) return forced;
    if (all_intra) return 0;
    if (rows >= H264E_RESIDENT_WG_V2 && parallel && !tree) return 4;
    if (wgs <= H264E_RESIDENT_WG_V3) return 3;
    return 2;
}

/* all members that are still in the group have submitted: merge and launch (g->mu held).  ONE launch per window geometry (narrow /
 * wide: different kernels), its variant chosen from the MERGED grid: a member's own few rows say nothing about how full the chip will
 * be, and the four-wave latency variant (512 resident workgroups) must never carry a grid that does not fit the chip. */
static int group_launch_locked(h264e_hip_group_t *g)
{
    int rc = 0;
    size_t need_tasks = 0, need_order = 0, task_off = 0, order_off = 0;
    g->err[0] = 0;
    g->timed[0] = g->timed[1] = 0;
    g->last_variant[0] = g->last_variant[1] = -1;
#define GFAIL(...) do { snprintf(g->err, sizeof(g->err), __VA_ARGS__); rc = -1; } while (0)
    if (hipSetDevice(g->device) != hipSuccess) GFAIL("group launch: hipSetDevice(%d) failed", g->device);
    /* buffers for the whole round, sized BEFORE the first launch: growing them between two launches of a round would free memory the first
     * one is still reading */
    for (int k = 0; k < g->nmembers; k++)
        if (g->pend[k]) { need_tasks += (size_t)g->pend_jobs[k]; need_order += (size_t)g->pend_jobs[k]*(size_t)(g->member[k]->G.nmby + 1); }
    if (!rc && need_tasks > g->tasks_cap)
    {
        if (g->tasks_dev) (void)hipFree(g->tasks_dev);           /* (synchronises with the previous round's launch, which every member has waited for anyway) */
        g->tasks_cap = need_tasks + 64;
        if (hipMalloc((void **)&g->tasks_dev, sizeof(h264e_frame_task_t)*g->tasks_cap) != hipSuccess) { g->tasks_dev = 0; g->tasks_cap = 0; GFAIL("group launch: device allocation failed"); }
    }
    if (!rc && need_order > g->order_cap)
    {
        if (g->order_dev) (void)hipFree(g->order_dev);
        g->order_cap = need_order + 4096;
        if (hipMalloc((void **)&g->order_dev, sizeof(uint32_t)*g->order_cap) != hipSuccess) { g->order_dev = 0; g->order_cap = 0; GFAIL("group launch: device allocation failed"); }
    }
    /* the merged launch owns the device like any other persistent launch (g_device_held): taken here, given back by the first member whose
     * h264e_hip_sync sees the launch drained, or when the group goes away */
    if (!rc && need_tasks && !g->holds_device) { device_token_take(g->device); g->holds_device = 1; }
    for (int narrow = 0; narrow < 2 && !rc; narrow++)
    {
        int idx[H264E_GROUP_MAX], n = 0, jobs = 0, forced = 0, all_intra = 1;
        for (int k = 0; k < g->nmembers; k++)
            if (g->pend[k] && g->pend_narrow[k] == narrow)
            {
                idx[n++] = k; jobs += g->pend_jobs[k];
                if (g->pend_waves[k] > 0) forced = g->pend_waves[k];     /* H264E_WAVES: the same for every pool of the process */
                if (g->pend_waves[k] != 0) all_intra = 0;               /* 0 = a launch of intra frames only, -1 = to be chosen here */
            }
        if (!n) continue;
        const h264e_geom_t &G = g->member[idx[0]]->G;
        const int rows = G.nmby + 1, lag = narrow ? H264E_NARROW_FRAME_LAG : H264E_FRAME_LAG;
        const size_t total = (size_t)jobs*rows;
        /* two or more streams in the grid are parallel work like slices are (a group of one: what h264e_hip_submit would decide) */
        int parallel = n >= 2 || G.nmby >= 200, tree = 0;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < g->pend_jobs[idx[i]]; j++)
            {
                const h264e_frame_task_t &t = g->pend_tasks[idx[i]][j];
                if (t.active && t.nslices >= 2) parallel = 1;
                if (t.active && t.walk_quiet) tree = 1;
            }
        const int variant = pick_variant(forced, all_intra, jobs*G.nmby, (int)total, parallel, tree);
        if (jobs >= 65536) { GFAIL("group launch: too many jobs"); break; }
        h264e_frame_task_t *th = (h264e_frame_task_t *)malloc(sizeof(h264e_frame_task_t)*(size_t)jobs);
        uint32_t *oh = (uint32_t *)malloc(sizeof(uint32_t)*total);
        if (!th || !oh) { free(th); free(oh); GFAIL("out of host memory"); break; }
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
        if (!start) { free(th); free(oh); GFAIL("out of host memory"); break; }
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
            if (hipEventRecord(p->ev_prep, p->stream) != hipSuccess || hipStreamWaitEvent(g->stream, p->ev_prep, 0) != hipSuccess) GFAIL("group launch: cannot order the launch behind a member's stream");
        }
        h264e_frame_task_t *td = g->tasks_dev + task_off;
        uint32_t *od = g->order_dev + order_off;
        if (!rc && (hipMemcpyAsync(td, th, sizeof(h264e_frame_task_t)*(size_t)jobs, hipMemcpyHostToDevice, g->stream) != hipSuccess ||
                    hipMemcpyAsync(od, oh, sizeof(uint32_t)*total, hipMemcpyHostToDevice, g->stream) != hipSuccess)) GFAIL("group launch: task upload failed");
        free(th); free(oh);            /* pageable sources: staged before the calls return */
        if (!rc)
        {
            (void)hipEventRecord(g->ev_t0[narrow], g->stream);
            bk_launch_mb(G, narrow, variant, jobs, (unsigned)total, td, od, g->stream);
            (void)hipEventRecord(g->ev_t1[narrow], g->stream);
            const hipError_t le = hipGetLastError();
            if (le != hipSuccess) GFAIL("group launch: %s", hipGetErrorString(le));
            else { g->timed[narrow] = 1; g->last_variant[narrow] = variant; }
        }
        task_off += (size_t)jobs; order_off += total;
    }
#undef GFAIL
    if (hipEventRecord(g->ev_done, g->stream) != hipSuccess && !rc) { snprintf(g->err, sizeof(g->err), "group launch: hipEventRecord failed"); rc = -1; }
    if (rc && !g->err[0]) snprintf(g->err, sizeof(g->err), "group launch failed");
    if (rc) snprintf(g_err, sizeof(g_err), "%s", g->err);
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
    if (rc) snprintf(g_err, sizeof(g_err), "%s", g->err[0] ? g->err : "group launch failed");       /* g_err is per thread: every member gets the text */
    pthread_mutex_unlock(&g->mu);
    return rc;
}

extern "C" int h264e_hip_group_join(h264e_hip_group_t *g, h264e_hip_pool_t *p)
{
    if (!g || !p || p->group) FAIL("group_join: bad argument");
    pthread_mutex_lock(&g->mu);
    /* how many streams one grid can hold: a far reference read waits for a workgroup up to (12 - lag) dispatch keys AHEAD of its own
     * (enc_kernels.h rv_wait_rect), i.e. about (12 - lag)*(nmby + 1)/2 workgroups per member stream, and all of those must be resident
     * together with the waiting one.  The variant is chosen from the merged grid (pick_variant): the two-wave kernels hold 1536 / 2048
     * workgroups -- 1400 with a margin --, and the four-wave latency variant (512) only ever carries a grid that is resident as a whole */
    const int window = (12 - H264E_NARROW_FRAME_LAG)*(p->G.nmby + 1)/2, room = 1400/(window > 0 ? window : 1);
    int bad = g->nmembers >= H264E_GROUP_MAX || (g->nmembers >= 1 && g->nmembers >= room) || p->device != g->device || g->arrived;
    if (!bad && g->nmembers)
    {
        const h264e_geom_t &A = g->member[0]->G, &B = p->G;
        bad = A.width != B.width || A.height != B.height || A.row_words != B.row_words || A.spin_limit != B.spin_limit;
    }
    if (!bad) { g->member[g->nmembers++] = p; p->group = g; }
    pthread_mutex_unlock(&g->mu);
    if (bad) FAIL("group_join: the group is full (at most %d streams of this picture size share a launch), busy, on another device or holds another picture size", room < H264E_GROUP_MAX ? room : H264E_GROUP_MAX);
    return 0;
}

extern "C" void h264e_hip_group_leave(h264e_hip_group_t *g, h264e_hip_pool_t *p)
{
    if (!g || !p || p->group != g) return;
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
        if (!g->nmembers && g->holds_device)
        {
            /* the last member is gone: nobody will sync the group's launches any more */
            (void)hipSetDevice(g->device);
            (void)hipStreamSynchronize(g->stream);
            g->holds_device = 0; device_token_give(g->device);
        }
    }
    p->group = 0;
    pthread_mutex_unlock(&g->mu);
}

extern "C" void h264e_hip_group_destroy(h264e_hip_group_t *g)
{
    if (!g) return;
    (void)hipSetDevice(g->device);
    (void)hipStreamSynchronize(g->stream);
    for (int k = 0; k < g->nmembers; k++) { g->member[k]->group = 0; free(g->pend_tasks[k]); }
    if (g->tasks_dev) (void)hipFree(g->tasks_dev);
    if (g->order_dev) (void)hipFree(g->order_dev);
    if (g->holds_device) { g->holds_device = 0; device_token_give(g->device); }
    (void)hipEventDestroy(g->ev_done);
    for (int k = 0; k < 2; k++) { (void)hipEventDestroy(g->ev_t0[k]); (void)hipEventDestroy(g->ev_t1[k]); }
    (void)hipStreamDestroy(g->stream);
    pthread_mutex_destroy(&g->mu); pthread_cond_destroy(&g->cv);
    free(g);
}

extern "C" int h264e_hip_submit(h264e_hip_pool_t *p, const h264e_hip_task_t *tasks)
{
    if (!p || !tasks) FAIL("submit: null argument");
    const h264e_geom_t &G = p->G;
    if (p->pending >= TASK_RING - 1 && h264e_hip_sync(p)) return -1;
    h264e_frame_task_t *host = (h264e_frame_task_t *)calloc((size_t)p->nchains, sizeof(h264e_frame_task_t));
    if (!host) FAIL("out of host memory");
    int any = 0, any_narrow = 0, any_wide = 0, njobs = 0, all_intra = 1, max_slices = 1, any_leaf = 0;
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
        if (t.stream_mode && t.walk_quiet) any_leaf = 1;
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
        d.mb_counter = p->mb_counter;
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
            /* the re-encode path is rare and synchronous: a blocking copy keeps the host array's lifetime simple */
            if (hipStreamSynchronize(p->stream) != hipSuccess || hipMemcpy(p->clu_dev[cs], t.mv_clusters_per_mb, n, hipMemcpyHostToDevice) != hipSuccess)
            {
                free(host);
                FAIL("mv_clusters upload failed");
            }
            d.clusters_per_mb = p->clu_dev[cs];
        }
        memcpy(d.qdat, t.qdat, sizeof(d.qdat));
        d.launch_id = launch_id;
        if (d.walk_on_device) p->traj_cur[t.slot] ^= 1;         /* this launch's walk writes the other buffer: it is the latest from now on */
    }
    if (!any) { free(host); return 0; }
    if (!p->group) device_acquire(p);   /* one launch at a time per device (see g_device_held); a launch group takes the token for its merged launch (group_launch_locked) */
    if (any_narrow && any_wide) { free(host); FAIL("submit: the jobs of one launch must agree on narrow_window"); }
    h264e_frame_task_t *slot = p->tasks_dev + (size_t)p->ring_pos*p->nchains;
    p->ring_pos = (p->ring_pos + 1) % TASK_RING;
    p->pending++;
    HIPCHK(hipSetDevice(p->device));
    /* pageable source: the runtime stages the copy before returning, so `host` can be freed right away */
    hipError_t e = hipMemcpyAsync(slot, host, sizeof(h264e_frame_task_t)*(size_t)p->nchains, hipMemcpyHostToDevice, p->stream);
    if (e != hipSuccess) { free(host); FAIL("task upload: %s", hipGetErrorString(e)); }
    e = hipMemsetAsync(p->progress_all, 0, sizeof(int)*2*(size_t)p->nchains*G.nmby, p->stream);
    /* rows kept from the previous encode of a frame count as complete */
    for (int c = 0; c < p->nchains && e == hipSuccess; c++)
        if (host[c].active && host[c].first_row > 0)
        {
            int *done = (int *)malloc(sizeof(int)*(size_t)host[c].first_row);
            if (!done) { e = hipErrorOutOfMemory; break; }
            for (int r = 0; r < host[c].first_row; r++) done[r] = G.nmbx + 1;
            e = hipMemcpyAsync(p->chains_host[host[c].chain].progress, done, sizeof(int)*(size_t)host[c].first_row, hipMemcpyHostToDevice, p->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(p->chains_host[host[c].chain].progress + G.nmby, done, sizeof(int)*(size_t)host[c].first_row, hipMemcpyHostToDevice, p->stream);
            free(done);         /* pageable source: staged before the call returns */
        }
    if (e != hipSuccess) { free(host); FAIL("progress reset: %s", hipGetErrorString(e)); }
    /* wavefronts per macroblock row: two (search | reconstruction pipeline) halve the macroblock latency for twice the wave slots --
     * the better trade wherever a launch is latency bound (single-slice streams: mis-speculation events; rate control and the
     * frame-at-a-time API: a few frames per launch) and still level for multi-slice streams; an all-intra launch has nothing to
     * search and no events: one wave per row, twice the rows in flight (22.4 vs 18.3 M MB/s at 1080p) */
    /* (waves = 0 selects the intra-only variant of the one-wave kernel: no inter code, half the registers, twice the rows in flight) */
    /* (... and 4 the two-wave kernel allocated for 4 waves per SIMD: launches bound by the rows in flight -- 8K-class pictures, many slices) */
    /* which launches that is: pick_variant above (re-measured in round 4 with the slimmer macroblock step) */
    /* (... and 3 the latency variant -- four waves per row: the 8x8 partition search and the deblocking + stores on waves of their own:
     * launches of one or a few frames, where the chip is empty and only the macroblock latency counts: the frame-at-a-time API) */
    const int waves = pick_variant(p->waves, all_intra, njobs*G.nmby, njobs*(G.nmby + 1), max_slices >= 2 || G.nmby >= 200, any_leaf);
    if (p->group)
    {
        /* member of a launch group: the launch is merged with the other members' and the variant is chosen from the MERGED grid
         * (group_launch_locked): this member only says what it cannot decide there -- forced by H264E_WAVES, intra frames only (0), or open (-1) */
        const int grc = group_submit(p, host, njobs, any_narrow, p->waves ? p->waves : all_intra ? 0 : -1);
        free(host);
        return grc;
    }
    free(host);
    /* the dispatch order for this launch's shape (jobs up to the last active one; window geometry) */
    if (njobs != p->order_jobs || any_narrow != p->order_narrow || (max_slices >= 2) != p->order_sliced)
    {
        if (build_order(p, njobs, any_narrow, max_slices >= 2)) FAIL("out of host memory");
        HIPCHK(hipMemcpyAsync(p->order, p->order_host, sizeof(uint32_t)*p->order_count, hipMemcpyHostToDevice, p->stream));     /* pageable: staged before the call returns */
        p->order_jobs = njobs; p->order_narrow = any_narrow; p->order_sliced = max_slices >= 2;
    }
    const int pe = p->ev_pending;
    if (p->profile) HIPCHK(hipEventRecord(p->ev[pe][0], p->stream));
    bk_launch_mb(G, any_narrow, waves, njobs, (unsigned)p->order_count, slot, p->order, p->stream);
    if (p->profile) HIPCHK(hipEventRecord(p->ev[pe][1], p->stream));
    HIPCHK(hipGetLastError());
    if (p->profile)
    {
        HIPCHK(hipEventRecord(p->ev[pe][2], p->stream));
        p->ev_pending++;
    }
    return 0;
}

extern "C" int h264e_hip_step_flags(h264e_hip_pool_t *p, int *flags /* [nchains][2] */)
{
    if (!p || !flags) FAIL("step_flags: bad argument");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpy(flags, p->stepflags, sizeof(int)*2*(size_t)p->nchains, hipMemcpyDeviceToHost));
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
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpy(dst, src, sizeof(int32_t)*2*(size_t)p->G.nmb, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int h264e_hip_stream_copy_picture(h264e_hip_pool_t *p, int from, int to)
{
    if (!p || from < 0 || to < 0 || from >= p->nchains || to >= p->nchains) FAIL("stream_copy_picture: bad argument");
    if (from == to) return 0;
    const size_t plane = (size_t)p->G.W*p->G.H*3/2;
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpyAsync(p->chains_host[to].rec[0][0], p->chains_host[from].rec[0][0], plane, hipMemcpyDeviceToDevice, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
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
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpyAsync(dst, p->chains_host[slot].nal_arena, nbytes, hipMemcpyDeviceToHost, p->copy_stream));
    HIPCHK(hipStreamSynchronize(p->copy_stream));
    return 0;
}

extern "C" int h264e_hip_download_i420(h264e_hip_pool_t *p, int first, int nframes, uint8_t *host)
{
    if (!p || !host || first < 0 || nframes < 0 || first + nframes > p->frames_resident) FAIL("download_i420: bad range");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpy(host, p->clip + p->frame_bytes*(size_t)first, p->frame_bytes*(size_t)nframes, hipMemcpyDeviceToHost));
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
    /* the kernel polls a word in device memory: the launch id is written there BY VALUE on a stream of its own, next to the running
     * launch -- not behind the application's staging uploads on the copy stream (up to hundreds of MB), and not as a copy whose source
     * could have moved on to the next launch's id by the time it executes */
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)p->abort_dev, p->launch_counter, 1, p->abort_stream));
    return 0;
}

/* 1 while work submitted to the pool is still running */
extern "C" int h264e_hip_busy(h264e_hip_pool_t *p)
{
    if (!p) return 0;
    (void)hipSetDevice(p->device);
    if (p->group && hipEventQuery(p->group->ev_done) == hipErrorNotReady) return 1;
    return hipStreamQuery(p->stream) == hipErrorNotReady;
}

extern "C" int h264e_hip_result(h264e_hip_pool_t *p, int chain, int slot, h264e_hip_result_t *res)
{
    if (!p || !res || chain < 0 || chain >= p->nchains || slot < 0 || slot >= p->slots) FAIL("result: bad argument");
    h264e_frameout_t f;
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpy(&f, p->chains_host[chain].fout + slot, sizeof(f), hipMemcpyDeviceToHost));
    res->nbytes = f.nbytes; res->all_skipped = f.all_skipped; res->clusters_moved = f.clusters_moved; res->overflow = f.overflow; res->far_reads = f.far_reads;
    res->nslices = f.nslices;
    for (int k = 0; k < H264E_HIP_MAX_SLICES; k++) res->slice_nbytes[k] = f.slice_nbytes[k];
    return 0;
}

extern "C" int h264e_hip_read_rbsp(h264e_hip_pool_t *p, int chain, int slot, uint8_t *dst, uint32_t cap)
{
    if (!p || !dst || chain < 0 || chain >= p->nchains || slot < 0 || slot >= p->slots) FAIL("read_rbsp: bad argument");
    h264e_frameout_t f;
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpy(&f, p->chains_host[chain].fout + slot, sizeof(f), hipMemcpyDeviceToHost));
    if (f.nbytes > cap) FAIL("read_rbsp: destination too small");
    HIPCHK(hipMemcpy(dst, p->chains_host[chain].arena + f.offset, f.nbytes, hipMemcpyDeviceToHost));
    return (int)f.nbytes;
}

extern "C" int h264e_hip_read_chain(h264e_hip_pool_t *p, int chain, int nslots, h264e_hip_result_t *res, uint32_t *offsets,
                                    uint8_t *arena_dst, uint32_t cap, uint32_t *used)
{
    if (!p || !res || !offsets || !arena_dst || chain < 0 || chain >= p->nchains || nslots < 0 || nslots > p->slots) FAIL("read_chain: bad argument");
    h264e_frameout_t *f = (h264e_frameout_t *)malloc(sizeof(h264e_frameout_t)*(size_t)(nslots ? nslots : 1));
    uint32_t cur = 0;
    if (!f) FAIL("out of host memory");
    if (hipSetDevice(p->device) != hipSuccess ||
        hipMemcpy(f, p->chains_host[chain].fout, sizeof(h264e_frameout_t)*(size_t)nslots, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(&cur, p->chains_host[chain].cursor, 4, hipMemcpyDeviceToHost) != hipSuccess)
    {
        free(f);
        FAIL("read_chain: copy failed");
    }
    for (int i = 0; i < nslots; i++)
    {
        res[i].nbytes = f[i].nbytes; res[i].all_skipped = f[i].all_skipped; res[i].clusters_moved = f[i].clusters_moved; res[i].overflow = f[i].overflow;
        res[i].far_reads = f[i].far_reads; res[i].nslices = f[i].nslices;
        for (int k = 0; k < H264E_HIP_MAX_SLICES; k++) res[i].slice_nbytes[k] = f[i].slice_nbytes[k];
        offsets[i] = f[i].offset;
    }
    free(f);
    if (cur > cap) FAIL("read_chain: destination too small (%u > %u)", cur, cap);
    HIPCHK(hipMemcpy(arena_dst, p->chains_host[chain].arena, cur, hipMemcpyDeviceToHost));
    if (used) *used = cur;
    return 0;
}

extern "C" int h264e_hip_read_mbrec(h264e_hip_pool_t *p, int chain, int slot, h264e_hip_mbrec_t *dst)
{
    if (!p || !dst || chain < 0 || chain >= p->nchains || slot < 0 || slot >= p->slots) FAIL("read_mbrec: bad argument");
    const size_t n = sizeof(h264e_mbrec_t)*(size_t)p->G.nmb;
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpy(dst, p->chains_host[chain].mbrec + (size_t)slot*p->G.nmb, n, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int h264e_hip_read_mbrec_all(h264e_hip_pool_t *p, int chain, int nslots, h264e_hip_mbrec_t *dst)
{
    if (!p || !dst || chain < 0 || chain >= p->nchains || nslots < 0 || nslots > p->slots) FAIL("read_mbrec_all: bad argument");
    const size_t n = sizeof(h264e_mbrec_t)*(size_t)p->G.nmb*(size_t)nslots;
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpy(dst, p->chains_host[chain].mbrec, n, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int h264e_hip_read_recon(h264e_hip_pool_t *p, int chain, uint8_t *dst)
{
    if (!p || !dst || chain < 0 || chain >= p->nchains) FAIL("read_recon: bad argument");
    const size_t n = (size_t)p->G.W*p->G.H*3/2;
    const uint8_t *src = p->chains_host[chain].rec[p->ref_sel[chain]][0];   /* after the swap: last reconstruction */
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpy(dst, src, n, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int h264e_hip_reset_results(h264e_hip_pool_t *p, int chain)
{
    if (!p || chain < 0 || chain >= p->nchains) FAIL("reset_results: bad argument");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemsetAsync(p->chains_host[chain].cursor, 0, 16, p->stream));
    return 0;
}

extern "C" int h264e_hip_rewind_frame(h264e_hip_pool_t *p, int chain, int slot)
{
    if (!p || chain < 0 || chain >= p->nchains || slot < 0 || slot >= p->slots) FAIL("rewind_frame: bad argument");
    p->ref_sel[chain] ^= 1;
    /* the frame's result is dropped too: the arena cursor goes back to where that result starts */
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpyAsync(p->chains_host[chain].cursor, &p->chains_host[chain].fout[slot].offset, sizeof(uint32_t), hipMemcpyDeviceToDevice, p->stream));
    return 0;
}

/* test hook: the dispatch order build_order makes for a launch of `jobs` jobs (banded = 1 forces the XCD bands with their padding, 0 none);
 * returns the number of entries (out gets at most cap of them), -1 on failure.  Leaves the pool's cached order invalid. */
extern "C" long h264e_hip_selftest_order(h264e_hip_pool_t *p, int jobs, int narrow, int banded, uint32_t *out, size_t cap)
{
    if (!p || jobs < 1 || jobs > p->nchains || !out) { snprintf(g_err, sizeof(g_err), "selftest_order: bad argument"); return -1; }
    const char *old = getenv("H264E_XCD_BANDS");
    char keep[32];
    if (old) snprintf(keep, sizeof(keep), "%s", old);
    setenv("H264E_XCD_BANDS", banded ? "8" : "0", 1);
    const int rc = build_order(p, jobs, narrow, 0);
    if (old) setenv("H264E_XCD_BANDS", keep, 1); else unsetenv("H264E_XCD_BANDS");
    p->order_jobs = -1;
    if (rc) { snprintf(g_err, sizeof(g_err), "out of host memory"); return -1; }
    memcpy(out, p->order_host, sizeof(uint32_t)*(p->order_count < cap ? p->order_count : cap));
    return (long)p->order_count;
}

extern "C" int h264e_hip_selftest_nal_escape(h264e_hip_pool_t *p, const uint8_t *src, uint32_t n, uint8_t *dst, uint32_t cap, uint32_t *out_n)
{
    if (!p || !src || !dst || !out_n) FAIL("selftest_nal_escape: bad argument");
    const size_t sb = ((size_t)n + 64 + 15) & ~(size_t)15, db = ((size_t)cap + 15) & ~(size_t)15;
    uint8_t *buf = 0;
    uint32_t res[2] = { 0, 0 };
    HIPCHK(hipSetDevice(p->device));
    if (hipMalloc((void **)&buf, sb + db + 64) != hipSuccess) FAIL("selftest_nal_escape: device allocation failed");
    hipError_t e = hipMemset(buf, 0, sb + db + 64);
    if (e == hipSuccess) e = hipMemcpy(buf, src, n, hipMemcpyHostToDevice);
    if (e == hipSuccess)
    {
        bk_launch_nal_selftest(buf + sb, cap, (const uint8_t *)buf, n, (uint32_t *)(buf + sb + db), p->stream);
        e = hipStreamSynchronize(p->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(res, buf + sb + db, sizeof(res), hipMemcpyDeviceToHost);
    if (e == hipSuccess && !res[1] && res[0] <= cap) e = hipMemcpy(dst, buf + sb, res[0], hipMemcpyDeviceToHost);
    (void)hipFree(buf);
    if (e != hipSuccess) FAIL("selftest_nal_escape: %s", hipGetErrorString(e));
    *out_n = res[0];
    return res[1] ? 1 : 0;
}

extern "C" int h264e_hip_selftest_stage(h264e_hip_pool_t *p, int stage, const uint8_t *in, uint32_t nin, const int *args /* [24] */, uint8_t *out, uint32_t nout)
{
    if (!p || !in || !args || !out || stage < 1 || stage > 8 || nin > STAGE_IN_MAX || nout > STAGE_OUT_MAX) FAIL("selftest_stage: bad argument");
    uint8_t *buf = 0;
    HIPCHK(hipSetDevice(p->device));
    if (hipMalloc((void **)&buf, STAGE_IN_MAX + STAGE_OUT_MAX + 256) != hipSuccess) FAIL("selftest_stage: device allocation failed");
    hipError_t e = hipMemset(buf, 0, STAGE_IN_MAX + STAGE_OUT_MAX + 256);
    if (e == hipSuccess) e = hipMemcpy(buf, in, nin, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(buf + STAGE_IN_MAX, args, STAGE_NARGS*sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess)
    {
        bk_launch_stage_selftest(stage, (const uint8_t *)buf, (const int *)(buf + STAGE_IN_MAX), buf + STAGE_IN_MAX + 128, p->stream);
        e = hipStreamSynchronize(p->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out, buf + STAGE_IN_MAX + 128, nout, hipMemcpyDeviceToHost);
    (void)hipFree(buf);
    if (e != hipSuccess) FAIL("selftest_stage: %s", hipGetErrorString(e));
    return 0;
}

/* diagnostic: per-phase cycle sums of the -DH264E_STAMPS build, summed over chains (zeros in the product build): [0..31] the rows'
 * phases, [32..47] the finalizer workgroups' (10 ns ticks of the constant 100 MHz clock) */
extern "C" int h264e_hip_stamps_read(h264e_hip_pool_t *p, unsigned long long *dst /* [48] */, int reset)
{
    if (!p || !dst) FAIL("stamps_read: bad argument");
    memset(dst, 0, 48*sizeof(unsigned long long));
    for (int c = 0; c < p->nchains; c++)
    {
        unsigned long long t[48];
        HIPCHK(hipSetDevice(p->device));
        HIPCHK(hipMemcpy(t, p->chains_host[c].prof, sizeof(t), hipMemcpyDeviceToHost));
        if (reset) HIPCHK(hipMemset(p->chains_host[c].prof, 0, sizeof(t)));
        for (int i = 0; i < 48; i++) dst[i] += t[i];
    }
    return 0;
}

/* macroblocks this pool's rows have reconstructed since the last reset -- everything the kernel worked on, including frames that a
 * mis-speculation or a rate-control miss threw away.  Call after h264e_hip_sync. */
extern "C" int h264e_hip_mb_counter(h264e_hip_pool_t *p, unsigned long long *count, int reset)
{
    if (!p || !count) FAIL("mb_counter: bad argument");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemcpy(count, p->mb_counter, sizeof(*count), hipMemcpyDeviceToHost));
    if (reset) HIPCHK(hipMemset(p->mb_counter, 0, sizeof(*count)));
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
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipEventRecord(p->ev_t0, p->stream));
    return 0;
}

extern "C" int h264e_hip_timer_stop(h264e_hip_pool_t *p, double *ms)
{
    if (!p || !ms) FAIL("timer_stop: null argument");
    *ms = 0;
    float f = 0;
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipEventRecord(p->ev_t1, p->stream));
    HIPCHK(hipEventSynchronize(p->ev_t1));
    HIPCHK(hipEventElapsedTime(&f, p->ev_t0, p->ev_t1));
    *ms = f;
    return 0;
}

#endif
