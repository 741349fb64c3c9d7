// Minimal reproducer for the "shift, clamp to 0..255, pack" pattern of the half-sample filters (enc_kernels.h halfpel3_win):
// does hipcc's fused form (v_ashr_pk_u8_i32 on gfx950, when it chooses it) give the same bytes as the unfused form and as the host?
// Prints one line per variant: <name> mismatches=<n> first=<index>.  Exit code 0 always; the test reads the lines.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __host__ static inline int clip255(int x) { return x < 0 ? 0 : x > 255 ? 255 : x; }
__device__ static inline int shr_opaque(int v, int s) { int t = v >> s; asm volatile("" : "+v"(t)); return t; }

template <bool OPAQUE> __global__ void pack_kernel(const int *in, unsigned *out, int n, int add, int sh)
{
    const int i = blockIdx.x*blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned o = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
        const int v = in[4*i + k] + add;
        const int f = OPAQUE ? clip255(shr_opaque(v, sh)) : clip255(v >> sh);
        o |= (unsigned)f << (8*k);
    }
    out[i] = o;
}

int main()
{
    const int n = 1 << 16;
    std::vector<int> h(4*n);
    unsigned seed = 12345;
    for (int i = 0; i < 4*n; i++)
    {
        seed = seed*1664525u + 1013904223u;
        const int r = (int)(seed >> 8);
        // ranges of the real filters: one-dimensional 6-tap sums -2550..10710, two-dimensional ones up to +-600000; plus edge values
        h[i] = (i % 7 == 0) ? (r % 1300001) - 650000 : (i % 7 == 1) ? (r % 64) - 32 : (i % 7 == 2) ? 8160 + (r % 64) - 32 : (r % 14001) - 3000;
    }
    int *din; unsigned *dout;
    if (hipMalloc(&din, sizeof(int)*4*n) != hipSuccess || hipMalloc(&dout, sizeof(unsigned)*n) != hipSuccess) { printf("no device\n"); return 0; }
    (void)hipMemcpy(din, h.data(), sizeof(int)*4*n, hipMemcpyHostToDevice);
    const int adds[2] = { 16, 512 }, shs[2] = { 5, 10 };
    for (int c = 0; c < 2; c++)
        for (int op = 0; op < 2; op++)
        {
            std::vector<unsigned> o(n);
            if (op) hipLaunchKernelGGL(pack_kernel<true>, dim3(n/256), dim3(256), 0, 0, din, dout, n, adds[c], shs[c]);
            else    hipLaunchKernelGGL(pack_kernel<false>, dim3(n/256), dim3(256), 0, 0, din, dout, n, adds[c], shs[c]);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(o.data(), dout, sizeof(unsigned)*n, hipMemcpyDeviceToHost);
            int bad = 0, first = -1, shown = 0;
            for (int i = 0; i < n; i++)
            {
                unsigned want = 0;
                for (int k = 0; k < 4; k++) want |= (unsigned)clip255((h[4*i + k] + adds[c]) >> shs[c]) << (8*k);
                if (want != o[i])
                {
                    if (first < 0) first = i;
                    bad++;
                    if (shown < 6)
                    {
                        printf("  detail %s_shift%d: inputs+add %d %d %d %d -> got %08x want %08x\n", op ? "opaque" : "plain", shs[c],
                               h[4*i] + adds[c], h[4*i + 1] + adds[c], h[4*i + 2] + adds[c], h[4*i + 3] + adds[c], o[i], want);
                        shown++;
                    }
                }
            }
            printf("%s_shift%d mismatches=%d first=%d\n", op ? "opaque" : "plain", shs[c], bad, first);
        }
    return 0;
}
