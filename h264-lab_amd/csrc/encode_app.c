/*
 * encode_app -- command-line encoder with the reference CLI's contract (/root/reference/src/minih264e_test.c,
 * "T:n" below), running on the MI355X HIP path through the H264E_* API.
 *
 * Kept from the reference: options are matched by prefix (T:121-131); EVERY --long option
 * consumes the following argv, flags included (T:212); -o / -i / -r short forms, -r overwriting the input name
 * (T:213-215); an unknown --option prints an error and parsing goes on (T:187-191), a bare argument prints the
 * error and exits 1 (T:219-223, T:476-477); picture size comes from the LAST WxH or size name inside the input
 * file name, default 352x288 (T:256-329, T:483-485); defaults gop 20, qp 33 (T:10-17); --maxframes only matters
 * when 0 (T:576); --kbps K -> desired_frame_bytes = K*1000/8/30, qp 10..50 (T:596-600); vbv_size_bytes = 12500
 * (T:524); --psnr makes const_input_flag 0 and prints the reference's PSNR line (T:331-405, T:521);
 * stdout carries "sizeof_persist = %d sizeof_scratch = %d" and, with --stats, "frame=%d, bytes=%d" (T:568, T:650).
 * --threads N (T:34-111, T:163) = N row-band slices per frame, the bitstream of the reference built with -DH264E_MAX_THREADS
 * (no host threads are involved here: the slices are wavefronts of the same kernel launch).
 * Not kept: --gen (libm-dependent synthetic input), --denoise: both are REFUSED (error line, exit 1, no output file).
 *
 * Extras (new option names, also argv-consuming): --device N; --clip 1 (THE DEFAULT) streams the file through the clip encoder
 * (H264E_clip_*: consecutive frames as a temporal wavefront on the GPU, same bitstream, bounded host memory whatever the file size;
 * --kbps, --threads, --psnr, --stats work there too), --clip 0 runs the reference's own loop over H264E_encode, one frame per call; --chains N bounds the frames in flight per launch; --gpus N cuts the file into N
 * GOP-aligned blocks, one clip encoder per block and GPU, with the mv_clusters state handed over and validated at every boundary.
 */
#define _FILE_OFFSET_BITS 64
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/h264e_mi355x.h"

static struct
{
    char input_file[1024], output_file[1024], recon_file[1024];
    int have_input, have_output;
    int gop, qp, kbps, max_frames, speed, stats, psnr, device, clip, chains, threads, gpus;
    int unsupported;                    /* a reference option this encoder refuses was given: no stream is written, exit 1 */
} cmd;

static int starts(const char *pattern, const char *p) { return !strncmp(pattern, p, strlen(pattern)); }

static void parse_long(const char *p, const char *val)
{
    const char *v = val ? val : "";
    /* T:135 --gen and T:163 --denoise are options of the reference this encoder does not implement: a stream that silently differs
     * from the reference's would be worse than none, so they end the run (main: exit 1) instead of being skipped like a typo */
    if (starts("gen", p) || starts("denoise", p))
    {
        printf("ERROR: option --%s is not supported by the MI355X encoder (reference-only: %s)\n", p,
               starts("gen", p) ? "libm-generated synthetic input" : "temporal denoiser");
        cmd.unsupported = 1;
    }
    else if (starts("gop", p)) cmd.gop = atoi(v);
    else if (starts("qp", p)) cmd.qp = atoi(v);
    else if (starts("kbps", p)) cmd.kbps = atoi(v);
    else if (starts("maxframes", p)) cmd.max_frames = atoi(v);
    else if (starts("speed", p)) cmd.speed = atoi(v);
    else if (starts("threads", p)) cmd.threads = atoi(v);       /* T:163-166: N row-band slices per frame (the reference's H264E_MAX_THREADS build) */
    else if (starts("stats", p)) cmd.stats = 1;
    else if (starts("psnr", p)) cmd.psnr = 1;
    else if (starts("output", p)) { snprintf(cmd.output_file, sizeof(cmd.output_file), "%s", v); cmd.have_output = 1; }
    else if (starts("input", p)) { snprintf(cmd.input_file, sizeof(cmd.input_file), "%s", v); cmd.have_input = 1; }
    else if (starts("recon", p)) snprintf(cmd.recon_file, sizeof(cmd.recon_file), "%s", v);
    else if (starts("device", p)) cmd.device = atoi(v);
    else if (starts("clip", p)) cmd.clip = atoi(v);
    else if (starts("chains", p)) cmd.chains = atoi(v);
    else if (starts("gpus", p)) cmd.gpus = atoi(v);
    else printf("ERROR: Unknown option %s\n", p);
}

static int read_cmdline(int argc, char **argv)
{
    int i;
    cmd.gop = 20; cmd.qp = 33; cmd.max_frames = 99999; cmd.device = -1; cmd.clip = -1;
    for (i = 1; i < argc; i++)
    {
        const char *p = argv[i];
        if (*p++ == '-')
        {
            const char *val = i + 1 < argc ? argv[i + 1] : NULL;
            switch (*p)
            {
            case '-': parse_long(p + 1, val); i++; break;
            case 'o': snprintf(cmd.output_file, sizeof(cmd.output_file), "%s", val ? val : ""); cmd.have_output = 1; i++; break;
            case 'i':
            case 'r': snprintf(cmd.input_file, sizeof(cmd.input_file), "%s", val ? val : ""); cmd.have_input = 1; i++; break;
            default: break;
            }
        } else
        {
            printf("ERROR: Unknown option %s\n", p);
            return 0;
        }
    }
    if (cmd.unsupported) return 0;
    if (!cmd.have_input)
    {
        printf("Usage:\n    encode_app [options] --input <input[frame_size].yuv> --output <output.264>\n"
               "Frame size can be: WxH sqcif qvga svga 4vga sxga xga vga qcif 4cif\n"
               "    4sif cif sif pal ntsc d1 16cif 16sif 720p 4SVGA 4XGA 16VGA 16VGA\n"
               "Options (every --option takes a value):\n"
               "    --input,  -i <f>  --output, -o <f>  --gop <n>  --qp <n>  --kbps <n>  --maxframes <n>\n"
               "    --speed <n>  --threads <n>  --stats x  --psnr x  --device <n>  --clip 0|1  --chains <n>  --gpus <n>\n");
        return 0;
    }
    return 1;
}

/* T:256-329 guess_format_from_name: the last WxH or size name in the string wins */
static void guess_format(const char *name, int *w, int *h)
{
    static const struct { const char *n; int w, h; } fmt[] = {
        { "sqcif", 128, 96 }, { "qvga", 320, 240 }, { "svga", 800, 600 }, { "4vga", 1280, 960 }, { "sxga", 1280, 1024 },
        { "xga", 1024, 768 }, { "vga", 640, 480 }, { "qcif", 176, 144 }, { "4cif", 704, 576 }, { "4sif", 704, 480 },
        { "cif", 352, 288 }, { "sif", 352, 240 }, { "pal", 720, 576 }, { "ntsc", 720, 480 }, { "d1", 720, 480 },
        { "16cif", 1408, 1152 }, { "16sif", 1408, 960 }, { "720p", 1280, 720 }, { "4SVGA", 1600, 1200 }, { "4XGA", 2048, 1536 },
        { "16VGA", 2560, 1920 }, { "16VGA", 2560, 1920 } };
    int i = (int)strlen(name), found = 0;
    while (--i >= 0)
    {
        const char *p = name + i;
        int prev = found;
        unsigned k;
        found = 0;
        if (*p >= '0' && *p <= '9')
        {
            char *end;
            int width = (int)strtoul(p, &end, 10);
            if (width && (*end == 'x' || *end == 'X') && (end[1] >= '1' && end[1] <= '9'))
            {
                int height = (int)strtoul(end + 1, &end, 10);
                if (height) { *w = width; *h = height; found = 1; }
            }
        }
        for (k = 0; k < sizeof(fmt)/sizeof(fmt[0]); k++)
            if (!strncmp(p, fmt[k].n, strlen(fmt[k].n))) { *w = fmt[k].w; *h = fmt[k].h; found = 1; }
        if (!found && prev) return;
    }
}

static struct { double noise[4], count[4], bytes; int frames; } g_psnr;

static void psnr_add(const unsigned char *p0, const unsigned char *p1, int w, int h, int bytes)   /* T:355-374 */
{
    int i, k;
    for (k = 0; k < 3; k++)
    {
        double s = 0;
        for (i = 0; i < w*h; i++) { int d = *p0++ - *p1++; s += d*d; }
        g_psnr.count[k] += w*h;
        g_psnr.noise[k] += s;
        if (!k) w >>= 1, h >>= 1;
    }
    g_psnr.frames++;
    g_psnr.bytes += bytes;
}

static void psnr_print(void)                                                                       /* T:376-405 */
{
    double fps = 30, kbps = g_psnr.bytes*8./((double)g_psnr.frames/(fps))/1000;
    double db = 10*log10(255.*255/(g_psnr.noise[0]/g_psnr.count[0]));
    int i;
    printf("%5.0f kbps@30fps  ", kbps);
    for (i = 0; i < 3; i++)
        printf(" %s=%.2f db ", i ? (i == 1 ? "UPSNR" : "VPSNR") : "YPSNR", 10*log10(255.*255/(g_psnr.noise[i]/g_psnr.count[i])));
    printf("  %6.2f db/rate ", 10*log10((double)g_psnr.count[0]*g_psnr.count[0]*3/2*255*255/(g_psnr.noise[0]*g_psnr.bytes)));
    printf("  %6.3f db/lgrate ", db/log10(kbps));
    printf("  \n");
}

/*
 * --clip 1: the file streams through the clip encoder in bounded memory (minih264e_test.c:584 / :654 read one frame, encode it,
 * write it -- here the same loop is a pipeline):
 *   reader thread : fread() into one of two pinned staging buffers (H264E_clip_host_alloc)
 *   copy engine   : staging buffer -> ring of input frames in HBM (H264E_clip_upload_async), issued and completed from the
 *                   encoder's idle hook, i.e. while earlier frames are being encoded
 *   GPU + caller  : H264E_clip_encode over the frames that have landed; its output buffer is flushed to the file when full
 * --psnr does not bring reconstructions back to the host: the sums of squared differences are taken on the device.
 */
#include <pthread.h>
#include <unistd.h>

typedef struct
{
    FILE *fin;
    size_t fsz;
    int nframes, chunk;                 /* frames still to read from the current file position, frames per staging buffer */
    int base;                           /* stream index (within the shard) of the first of them */
    uint8_t *buf[2];
    int have[2];                        /* frames waiting in buf[k] (0 = free) */
    int first[2];                       /* their first frame index */
    int eof, error, quit;
    pthread_mutex_t mu;
    pthread_cond_t cv;
} feeder_t;

static void *feeder_thread(void *arg)
{
    feeder_t *f = (feeder_t *)arg;
    int next = 0, k = 0;
    while (next < f->nframes)
    {
        const int n = f->nframes - next < f->chunk ? f->nframes - next : f->chunk;
        pthread_mutex_lock(&f->mu);
        while (f->have[k] && !f->quit) pthread_cond_wait(&f->cv, &f->mu);
        pthread_mutex_unlock(&f->mu);
        if (f->quit) break;
        if (fread(f->buf[k], f->fsz, (size_t)n, f->fin) != (size_t)n) { f->error = 1; break; }
        pthread_mutex_lock(&f->mu);
        f->first[k] = f->base + next; f->have[k] = n;
        pthread_cond_broadcast(&f->cv);
        pthread_mutex_unlock(&f->mu);
        next += n; k ^= 1;
    }
    pthread_mutex_lock(&f->mu);
    f->eof = 1;
    pthread_cond_broadcast(&f->cv);
    pthread_mutex_unlock(&f->mu);
    return NULL;
}

typedef struct { feeder_t *f; H264E_clip_t *clip; int resident, inflight /* buffer being copied, or -1 */, next_buf; int failed /* an upload was refused or lost: H264E_last_error() says why */; } pump_t;

/* idle hook: finish the copy in flight, start the next one if a buffer is ready and the input ring has room */
static void pump(void *token)
{
    pump_t *p = (pump_t *)token;
    feeder_t *f = p->f;
    int st;
    if (p->failed) return;
    st = p->inflight >= 0 ? H264E_clip_upload_poll(p->clip) : 0;
    if (st < 0) { p->failed = 1; return; }              /* the copy stream reported an error (lost device, ...) */
    if (st == 1)
    {
        pthread_mutex_lock(&f->mu);
        f->have[p->inflight] = 0;
        pthread_cond_broadcast(&f->cv);
        pthread_mutex_unlock(&f->mu);
        p->inflight = -1;
    }
    if (p->inflight < 0)
    {
        int n, first, next_frame;
        pthread_mutex_lock(&f->mu);
        n = f->have[p->next_buf]; first = f->first[p->next_buf];
        pthread_mutex_unlock(&f->mu);
        H264E_clip_position(p->clip, &next_frame, NULL);
        if (n && first + n - p->resident <= next_frame)
        {
            if (H264E_clip_upload_async(p->clip, first, n, f->buf[p->next_buf])) { p->failed = 1; return; }
            p->inflight = p->next_buf;
            p->next_buf ^= 1;
        }
    }
}

/* one shard = a GOP-aligned block of the file on one GPU (the whole file without --gpus) */
typedef struct
{
    int device, file_first, nframes, w, h;
    FILE *fout;                         /* write the stream as it is produced (single shard), or NULL: collect it in `mem` */
    uint8_t *mem; size_t mem_len, mem_cap;
    size_t *foff;                       /* [nframes + 1] byte offset of every frame in the shard's stream */
    int *fsize;                         /* [nframes] */
    uint64_t *ssd;                      /* [nframes][3] or NULL */
    H264E_clip_t *clip;
    H264E_clip_param_t par;
    uint8_t *stage[2], *out;
    size_t out_cap;
    int chunk, chains, rounds, relaunches, reencoded, rc;
    double enc_ms;
    pthread_t thread;
} shard_t;

static int shard_open(shard_t *s)
{
    const size_t fsz = (size_t)s->w*s->h*3/2;
    /* budgets (host: two staging buffers + the output buffer; HBM: the input ring); the environment overrides exist for tests that
     * want the ring to wrap and the output buffer to fill on tiny clips */
    const size_t stage_bytes = getenv("H264E_APP_STAGE_KB") ? (size_t)atol(getenv("H264E_APP_STAGE_KB")) << 10 : (size_t)384 << 20;
    const size_t ring_bytes = getenv("H264E_APP_RING_KB") ? (size_t)atol(getenv("H264E_APP_RING_KB")) << 10 : (size_t)6 << 30;
    const size_t out_cap = getenv("H264E_APP_OUT_KB") ? (size_t)atol(getenv("H264E_APP_OUT_KB")) << 10 : (size_t)64 << 20;
    const int n = s->nframes;
    s->chunk = (int)(stage_bytes/fsz); if (s->chunk < 1) s->chunk = 1; if (s->chunk > n) s->chunk = n;
    s->par.resident_frames = (int)(ring_bytes/fsz);
    if (s->par.resident_frames < 4*s->chunk) s->par.resident_frames = 4*s->chunk;
    if (s->par.resident_frames > n) s->par.resident_frames = n;
    s->out_cap = out_cap > 2*fsz + (1 << 16) ? out_cap : 2*fsz + (1 << 16);
    s->stage[0] = (uint8_t *)H264E_clip_host_alloc(fsz*(size_t)s->chunk);
    s->stage[1] = (uint8_t *)H264E_clip_host_alloc(fsz*(size_t)s->chunk);
    s->out = (uint8_t *)malloc(s->out_cap);
    s->fsize = (int *)calloc((size_t)n, sizeof(int));
    s->foff = (size_t *)calloc((size_t)n + 1, sizeof(size_t));
    if (cmd.psnr) s->ssd = (uint64_t *)calloc(3*(size_t)n, sizeof(uint64_t));
    if (!s->stage[0] || !s->stage[1] || !s->out || !s->fsize || !s->foff || (cmd.psnr && !s->ssd)) { printf("ERROR: not enough memory\n"); return 1; }
    if (H264E_clip_open(&s->clip, &s->par, n)) { printf("ERROR: %s\n", H264E_last_error()); return 1; }
    return 0;
}

static void shard_close(shard_t *s)
{
    if (s->clip) { (void)H264E_clip_upload_wait(s->clip); H264E_clip_close(s->clip); }
    H264E_clip_host_free(s->stage[0]); H264E_clip_host_free(s->stage[1]);
    free(s->out); free(s->fsize); free(s->foff); free(s->ssd); free(s->mem);
}

/* (re)encode frames [from, nframes) of the shard: reader thread -> staging buffers -> HBM ring -> clip encoder -> output */
static int shard_encode_from(shard_t *s, int from)
{
    const size_t fsz = (size_t)s->w*s->h*3/2;
    feeder_t fd;
    pump_t pp;
    pthread_t th;
    H264E_clip_stats_t st;
    uint64_t *ssd_tmp = s->ssd ? (uint64_t *)malloc(sizeof(uint64_t)*3*(size_t)s->nframes) : NULL;
    int *sizes = (int *)malloc(sizeof(int)*(size_t)s->nframes);
    int done = from, rc = 1, i;
    FILE *fin = fopen(cmd.input_file, "rb");
    if (!fin || !sizes || (s->ssd && !ssd_tmp)) { printf("ERROR: cant open input file %s\n", cmd.input_file); return 1; }
    fseeko(fin, (off_t)fsz*(off_t)(s->file_first + from), SEEK_SET);
    memset(&fd, 0, sizeof(fd));
    fd.fin = fin; fd.fsz = fsz; fd.nframes = s->nframes - from; fd.base = from; fd.chunk = s->chunk;
    fd.buf[0] = s->stage[0]; fd.buf[1] = s->stage[1];
    pthread_mutex_init(&fd.mu, NULL); pthread_cond_init(&fd.cv, NULL);
    pp.f = &fd; pp.clip = s->clip; pp.resident = s->par.resident_frames; pp.inflight = -1; pp.next_buf = 0; pp.failed = 0;
    H264E_clip_set_idle_hook(s->clip, pump, &pp);
    s->mem_len = s->foff[from];
    if (pthread_create(&th, NULL, feeder_thread, &fd)) { printf("ERROR: cannot start the reader thread\n"); fclose(fin); return 1; }
    while (done < s->nframes)
    {
        int next_frame, avail;
        size_t nb = 0, pos = 0;
        pump(&pp);
        H264E_clip_position(s->clip, &next_frame, &avail);
        if (pp.failed) { printf("ERROR: upload failed: %s\n", H264E_last_error()); goto out; }
        if (avail <= next_frame)
        {
            int starved;
            if (fd.error) { printf("ERROR: short read\n"); goto out; }
            /* nothing to encode yet: fine while the reader or a copy is still at work -- but when the reader has finished, no
             * buffer holds frames and no copy is in flight, nothing will ever arrive */
            pthread_mutex_lock(&fd.mu);
            starved = fd.eof && !fd.have[0] && !fd.have[1] && pp.inflight < 0;
            pthread_mutex_unlock(&fd.mu);
            if (starved) { printf("ERROR: input ended after %d of %d frames\n", avail, s->nframes); goto out; }
            usleep(200);
            continue;
        }
        if (ssd_tmp) H264E_clip_set_ssd_output(s->clip, ssd_tmp);
        if (H264E_clip_encode(s->clip, s->out, s->out_cap, &nb, sizes, 0, &st)) { printf("ERROR: %s\n", H264E_last_error()); goto out; }
        if (s->fout) { if (nb && !fwrite(s->out, nb, 1, s->fout)) { printf("ERROR writing output file\n"); goto out; } }
        else
        {
            if (s->mem_len + nb > s->mem_cap)
            {
                const size_t ncap = (s->mem_len + nb)*2 + (1 << 20);
                uint8_t *t = (uint8_t *)realloc(s->mem, ncap);
                if (!t) { printf("ERROR: not enough memory\n"); goto out; }
                s->mem = t; s->mem_cap = ncap;
            }
            memcpy(s->mem + s->mem_len, s->out, nb);
        }
        for (i = 0; i < st.frames; i++)
        {
            const int f = st.first_frame + i;
            s->fsize[f] = sizes[i];
            s->foff[f] = s->mem_len + pos;
            pos += (size_t)sizes[i];
            s->foff[f + 1] = s->mem_len + pos;
            if (ssd_tmp) memcpy(s->ssd + 3*(size_t)f, ssd_tmp + 3*(size_t)i, 3*sizeof(uint64_t));
        }
        s->mem_len += nb;
        done += st.frames; s->rounds += st.rounds; s->relaunches += st.reencoded_gops; s->enc_ms += st.encode_ms; s->chains = st.chains;
    }
    rc = 0;
out:
    pthread_mutex_lock(&fd.mu);
    fd.quit = 1; fd.have[0] = fd.have[1] = 0;
    pthread_cond_broadcast(&fd.cv);
    pthread_mutex_unlock(&fd.mu);
    pthread_join(th, NULL);
    (void)H264E_clip_upload_wait(s->clip);
    H264E_clip_set_idle_hook(s->clip, NULL, NULL);
    fclose(fin);
    free(sizes); free(ssd_tmp);
    return rc;
}

static void *shard_thread(void *arg)
{
    shard_t *s = (shard_t *)arg;
    s->rc = shard_encode_from(s, 0);
    return NULL;
}

/*
 * --clip 1 [--gpus N]: without --gpus one shard covers the file and writes the stream as it is produced.  With --gpus N the
 * file is cut into N contiguous GOP-aligned blocks, one clip encoder per block on device k (minih264e_test.c:576-662 is the
 * loop this replaces).  A block starts from a SPECULATED mv_clusters state; once the block in front of
 * it is final, its exact end state is handed over (8 bytes) and H264E_clip_revalidate names the GOP from which the block has
 * to be encoded again, if any; idr_pic_id parity is computed.  The blocks' bytes are written in order at the end.
 */
static int run_clip_mode(FILE *fout, int w, int h, long long total)
{
    const size_t fsz = (size_t)w*h*3/2;
    const int n = (int)((unsigned long long)total/fsz), gop = cmd.gop > 0 ? cmd.gop : n > 0 ? n : 1;
    const int ngop = (n + gop - 1)/gop;
    int nsh = cmd.gpus > 1 ? cmd.gpus : 1, k, g0 = 0, rc = 1, ndev = cmd.gpus > 1 ? H264E_device_count() : 1, f;
    shard_t *sh;
    int32_t state[2] = { 0, 0 };
    if (n <= 0) return 0;
    if (cmd.kbps) nsh = 1;                  /* rate control state crosses every frame: one shard */
    if (nsh > ngop) nsh = ngop;
    if (ndev < 1) ndev = 1;
    sh = (shard_t *)calloc((size_t)nsh, sizeof(*sh));
    if (!sh) return 1;
    for (k = 0; k < nsh; k++)
    {
        const int g1 = g0 + (ngop - g0)/(nsh - k);
        shard_t *s = sh + k;
        s->w = w; s->h = h;
        s->file_first = g0*gop; s->nframes = (g1*gop < n ? g1*gop : n) - g0*gop;
        s->device = cmd.device >= 0 && nsh == 1 ? cmd.device : k % ndev;
        s->fout = nsh == 1 ? fout : NULL;
        s->par.width = w; s->par.height = h; s->par.gop = cmd.gop; s->par.qp = cmd.qp; s->par.speed = cmd.speed; s->par.vbv_size_bytes = 100000/8;
        s->par.device = s->device; s->par.max_chains = cmd.chains; s->par.slices = cmd.threads; s->par.kbps = cmd.kbps;
        s->par.first_idr_pic_id_state = g0 & 1;         /* idr_pic_id toggles with every key frame (h264-lab.h:6774) */
        s->par.keep_records = nsh > 1;
        if (shard_open(s)) goto out;
        g0 = g1;
    }
    if (nsh == 1) sh[0].rc = shard_encode_from(sh, 0);
    else
    {
        for (k = 0; k < nsh; k++) if (pthread_create(&sh[k].thread, NULL, shard_thread, sh + k)) { printf("ERROR: cannot start a shard thread\n"); goto out; }
        for (k = 0; k < nsh; k++) pthread_join(sh[k].thread, NULL);
    }
    for (k = 0; k < nsh; k++) if (sh[k].rc) goto out;
    /* settle the shards in stream order: exact state in, encode again from the first GOP that consumed different candidates */
    for (k = 1; k < nsh; k++)
    {
        int32_t rs[2], es[2];
        int rf;
        if (k == 1 && H264E_clip_revalidate(sh[0].clip, state, &rf, rs, state)) goto out;      /* shard 0 started exact: its end state */
        for (;;)
        {
            if (H264E_clip_revalidate(sh[k].clip, state, &rf, rs, es)) { printf("ERROR: revalidation failed\n"); goto out; }
            if (rf < 0) break;
            sh[k].reencoded += sh[k].nframes - rf;
            if (H264E_clip_restart(sh[k].clip, rf, rs) || shard_encode_from(sh + k, rf)) goto out;
        }
        state[0] = es[0]; state[1] = es[1];
    }
    for (k = 0; k < nsh && nsh > 1; k++)
        if (sh[k].mem_len && !fwrite(sh[k].mem, sh[k].mem_len, 1, fout)) { printf("ERROR writing output file\n"); goto out; }
    for (k = 0, f = 0; k < nsh; k++)
    {
        int i;
        for (i = 0; i < sh[k].nframes; i++, f++)
        {
            if (cmd.stats) printf("frame=%d, bytes=%d\n", f, sh[k].fsize[i]);
            if (sh[k].ssd)
            {
                int c, pw = w, ph = h;
                for (c = 0; c < 3; c++)
                {
                    g_psnr.count[c] += pw*ph;
                    g_psnr.noise[c] += (double)sh[k].ssd[3*(size_t)i + c];
                    if (!c) pw >>= 1, ph >>= 1;
                }
                g_psnr.frames++;
                g_psnr.bytes += sh[k].fsize[i];
            }
        }
    }
    if (cmd.psnr) psnr_print();
    for (k = 0; k < nsh; k++)
        fprintf(stderr, "clip%s: %d frames%s, %d in flight per launch, %d launches (%d after a mis-speculated mv_clusters state), encode %.1f ms, input ring %d frames, staging 2 x %d frames%s\n",
                nsh > 1 ? " shard" : "", sh[k].nframes, nsh > 1 ? " on its GPU" : "", sh[k].chains, sh[k].rounds, sh[k].relaunches, sh[k].enc_ms,
                sh[k].par.resident_frames, sh[k].chunk, sh[k].reencoded ? " -- frames encoded again after the hand-off of the exact state" : "");
    if (nsh > 1)
    {
        int redo = 0;
        for (k = 0; k < nsh; k++) redo += sh[k].reencoded;
        fprintf(stderr, "stream of %d frames GOP-sharded over %d clip encoders on %d device(s): %d frames encoded again after state hand-off\n", n, nsh, ndev < nsh ? ndev : nsh, redo);
    }
    rc = 0;
out:
    for (k = 0; k < nsh; k++) shard_close(sh + k);
    free(sh);
    return rc;
}

int main(int argc, char **argv)
{
    H264E_create_param_t create_param;
    H264E_run_param_t run_param;
    H264E_io_yuv_t yuv;
    H264E_persist_t *enc;
    H264E_scratch_t *scratch;
    FILE *fin, *fout;
    uint8_t *buf_in, *buf_save, *coded_data;
    int w = 352, h = 288, i, frames = 0, frame_size, sizeof_persist = 0, sizeof_scratch = 0, error, sizeof_coded_data;

    if (!read_cmdline(argc, argv)) return 1;
    guess_format(cmd.input_file, &w, &h);
    fin = fopen(cmd.input_file, "rb");
    if (!fin) { printf("ERROR: cant open input file %s\n", cmd.input_file); return 1; }
    fout = fopen(cmd.have_output ? cmd.output_file : "out.264", "wb");
    if (!fout) { printf("ERROR: cant open output file %s\n", cmd.output_file); return 1; }
    if (cmd.device >= 0) H264E_set_device(cmd.device);

    memset(&create_param, 0, sizeof(create_param));
    memset(&run_param, 0, sizeof(run_param));
    create_param.enableNEON = 1;
    create_param.num_layers = 1;
    create_param.gop = cmd.gop;
    create_param.height = h;
    create_param.width = w;
    create_param.const_input_flag = cmd.psnr ? 0 : 1;
    create_param.vbv_size_bytes = 100000/8;

    error = H264E_sizeof(&create_param, &sizeof_persist, &sizeof_scratch);
    if (error) { printf("H264E_init error = %d\n", error); return 0; }
    printf("sizeof_persist = %d sizeof_scratch = %d\n", sizeof_persist, sizeof_scratch);

    /* the file goes through the streaming clip encoder unless --clip 0 asks for the reference's frame-at-a-time loop (same bytes,
     * one pipeline latency per frame) */
    if ((cmd.clip != 0 && cmd.max_frames) || cmd.gpus > 1)
    {
        /* the clip pipeline wants the frame count up front; an input that cannot say (a pipe, /dev/stdin) goes through the
         * reference's own read-until-EOF loop below instead (T:584), and an empty file gives an empty stream like there (T:654) */
        const long long total = fseeko(fin, 0, SEEK_END) ? -1 : (long long)ftello(fin);
        if (total >= 0)
        {
            const int r = run_clip_mode(fout, w, h, total);
            fclose(fin); fclose(fout);
            return r;
        }
        if (cmd.gpus > 1) { printf("ERROR: --gpus needs a seekable input file\n"); return 1; }
        clearerr(fin);
    }

    frame_size = w*h*3/2;
    buf_in = (uint8_t *)malloc((size_t)frame_size);
    buf_save = (uint8_t *)malloc((size_t)frame_size);
    enc = (H264E_persist_t *)malloc((size_t)sizeof_persist);
    scratch = (H264E_scratch_t *)malloc((size_t)sizeof_scratch);
    if (!buf_in || !buf_save || !enc || !scratch) { printf("ERROR: not enough memory\n"); return 1; }
    error = H264E_init(enc, &create_param);
    if (error) { printf("H264E_init error = %d (%s)\n", error, H264E_last_error()); return 1; }
    if (cmd.threads > 1 && H264E_set_slices(enc, cmd.threads)) { printf("ERROR: --threads %d not supported\n", cmd.threads); return 1; }

    for (i = 0; cmd.max_frames; i++)
    {
        if (!fread(buf_in, (size_t)frame_size, 1, fin)) break;
        if (cmd.psnr) memcpy(buf_save, buf_in, (size_t)frame_size);
        yuv.yuv[0] = buf_in; yuv.stride[0] = w;
        yuv.yuv[1] = buf_in + w*h; yuv.stride[1] = w/2;
        yuv.yuv[2] = buf_in + w*h*5/4; yuv.stride[2] = w/2;
        run_param.frame_type = 0;
        run_param.encode_speed = cmd.speed;
        if (cmd.kbps)
        {
            run_param.desired_frame_bytes = cmd.kbps*1000/8/30;
            run_param.qp_min = 10;
            run_param.qp_max = 50;
        } else
            run_param.qp_min = run_param.qp_max = cmd.qp;
        error = H264E_encode(enc, scratch, &run_param, &yuv, &coded_data, &sizeof_coded_data);
        if (error) { printf("H264E_encode error = %d (%s)\n", error, H264E_last_error()); return 1; }
        if (cmd.stats) printf("frame=%d, bytes=%d\n", frames++, sizeof_coded_data);
        if (!fwrite(coded_data, (size_t)sizeof_coded_data, 1, fout)) { printf("ERROR writing output file\n"); break; }
        if (cmd.psnr) psnr_add(buf_save, buf_in, w, h, sizeof_coded_data);
    }
    if (cmd.psnr) psnr_print();
    H264E_close(enc);
    free(enc); free(scratch); free(buf_in); free(buf_save);
    fclose(fin); fclose(fout);
    return 0;
}
