/*
 * oracle_app -- command-line front end of the CPU oracle (TEST INFRASTRUCTURE).
 * Accepts the subset of the reference encode_app options the oracle models
 * (/root/reference/src/minih264e_test.c:133-224): --input --output --qp --gop --speed --kbps --threads (row-band slices of the -DH264E_MAX_THREADS build).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "h264o.h"

int main(int argc, char **argv)
{
    const char *in = NULL, *out = "out.264";
    h264o_param_t par;
    int i, w = 352, h = 288, n = 0;
    FILE *fi, *fo;
    uint8_t *buf;
    h264o_enc_t *e;
    memset(&par, 0, sizeof(par));
    par.gop = 20; par.qp = 33; par.vbv_size_bytes = 100000/8;
    for (i = 1; i + 1 < argc; i += 2)
    {
        if (!strcmp(argv[i], "--input")) in = argv[i + 1];
        else if (!strcmp(argv[i], "--output")) out = argv[i + 1];
        else if (!strcmp(argv[i], "--qp")) par.qp = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--gop")) par.gop = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--speed")) par.speed = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--kbps")) par.kbps = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--threads")) par.slices = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--size")) sscanf(argv[i + 1], "%dx%d", &w, &h);
        else if (!strcmp(argv[i], "--frames")) n = atoi(argv[i + 1]);
    }
    if (!in) { fprintf(stderr, "usage: oracle_app --input f_WxH.yuv --output f.264 [--qp n --gop n --speed n --size WxH]\n"); return 1; }
    {
        const char *p = in + strlen(in);
        while (p > in) { int a, b; p--; if (sscanf(p, "%dx%d", &a, &b) == 2 && (p == in || p[-1] < '0' || p[-1] > '9')) { w = a; h = b; break; } }
    }
    par.width = w; par.height = h;
    fi = fopen(in, "rb"); fo = fopen(out, "wb");
    if (!fi || !fo) { fprintf(stderr, "cannot open files\n"); return 1; }
    buf = (uint8_t *)malloc((size_t)w*h*3/2);
    e = h264o_open(&par);
    for (i = 0; (!n || i < n) && fread(buf, (size_t)w*h*3/2, 1, fi) == 1; i++)
    {
        const uint8_t *yuv[3] = { buf, buf + w*h, buf + w*h*5/4 };
        int stride[3] = { w, w/2, w/2 }, nb;
        uint8_t *p;
        h264o_encode(e, yuv, stride, &p, &nb);
        fwrite(p, (size_t)nb, 1, fo);
    }
    h264o_close(e);
    free(buf);
    fclose(fi); fclose(fo);
    return 0;
}
