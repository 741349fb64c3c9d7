/*
 * vbv_harness.c -- TEST INFRASTRUCTURE.  Generator of tests/golden/vbv.json: streams of the REFERENCE encoder under frame-level rate
 * control with H264E_set_vbv_state (h264-lab.h:6898-6913) called in the middle of the stream -- the reference CLI never calls it, so
 * its binary cannot produce these.  Compiles the reference header into this translation unit and only CALLS its public
 * API the way minih264e_test.c:507-526, :592-604 sets it up; the input is the synth_v1 clip of oracle/synth_v1.c.  Built and run in the
 * build container only (`make -C oracle vbv`, tests/golden/make_golden_vbv.py); the stream's md5 and frame sizes -- data -- are committed.
 *
 *   vbv_harness W H NFRAMES GOP KBPS out.264 [frame:vbv_size_bytes:vbv_fullness_bytes ...]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#define MINIH264_IMPLEMENTATION
#include "h264-lab.h"
#include "synth_v1.h"

int main(int argc, char **argv)
{
    H264E_create_param_t cp;
    H264E_run_param_t rp;
    H264E_io_yuv_t io;
    H264E_persist_t *enc;
    H264E_scratch_t *scratch;
    int sp = 0, ss = 0, w, h, n, gop, kbps, t, k;
    uint8_t *frame;
    FILE *f;
    if (argc < 7) { fprintf(stderr, "usage: vbv_harness W H NFRAMES GOP KBPS out.264 [frame:size:fullness ...]\n"); return 2; }
    w = atoi(argv[1]); h = atoi(argv[2]); n = atoi(argv[3]); gop = atoi(argv[4]); kbps = atoi(argv[5]);
    memset(&cp, 0, sizeof(cp));
    cp.width = w; cp.height = h; cp.gop = gop; cp.vbv_size_bytes = 100000/8; cp.const_input_flag = 1; cp.enableNEON = 1; cp.num_layers = 1;
    if (H264E_sizeof(&cp, &sp, &ss)) return 1;
    enc = (H264E_persist_t *)malloc((size_t)sp + 64);
    scratch = (H264E_scratch_t *)malloc((size_t)ss + 64);
    frame = (uint8_t *)malloc((size_t)w*h*3/2);
    if (!enc || !scratch || !frame || H264E_init(enc, &cp)) return 1;
    f = fopen(argv[6], "wb");
    if (!f) return 1;
    for (t = 0; t < n; t++)
    {
        unsigned char *coded = NULL;
        int bytes = 0;
        for (k = 7; k < argc; k++)
        {
            int at, size, full;
            if (sscanf(argv[k], "%d:%d:%d", &at, &size, &full) == 3 && at == t) H264E_set_vbv_state(enc, size, full);
        }
        synth_v1_frame(frame, w, h, t, 1);
        io.yuv[0] = frame; io.yuv[1] = frame + (size_t)w*h; io.yuv[2] = io.yuv[1] + (size_t)(w/2)*(h/2);
        io.stride[0] = w; io.stride[1] = io.stride[2] = w/2;
        memset(&rp, 0, sizeof(rp));
        rp.frame_type = 0; rp.encode_speed = 0;
        rp.desired_frame_bytes = kbps*1000/8/30; rp.qp_min = 10; rp.qp_max = 50;
        if (H264E_encode(enc, scratch, &rp, &io, &coded, &bytes)) return 1;
        fwrite(coded, 1, (size_t)bytes, f);
        printf("frame=%d, bytes=%d\n", t, bytes);
    }
    fclose(f);
    return 0;
}
