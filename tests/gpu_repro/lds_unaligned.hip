// Does the MI355X serve LDS dword / qword reads at ANY byte address (SH_MEM_CONFIG alignment mode "unaligned", which the ROCm kernel driver
// selects for gfx9-class parts), and does it do so when the access straddles a bank, a 64-byte line, and between lanes that hit the same
// dwords?  The encoder's reference window is read at arbitrary sample positions (enc_kernels.h lds32u: two aligned reads + v_alignbyte per
// four samples); one unaligned ds_read_b32 would do.  hipcc does not assume the capability for gfx950 (it splits an align-1 LDS load into
// byte reads), so the read is forced with inline assembly.  Prints one line per variant: <name> mismatches=<n> first=<index>.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

// the forms the compiler may merge neighbouring dword reads into: ds_read2_b32 (two dwords at ADDR + 4*off), ds_read_b128
__global__ void __launch_bounds__(64) probe_wide(const unsigned char *src, unsigned *out, int stride, int n)
{
    __shared__ __attribute__((aligned(16))) unsigned char buf[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) buf[i] = src[i];
    __syncthreads();
    for (int base = 0; base < n; base += 64)
    {
        const int k = base + (int)threadIdx.x;
        const unsigned addr = (unsigned)(size_t)(buf) + (unsigned)((k*stride) % 7900);
        unsigned long long a; unsigned b0, b1, b2, b3;
        asm volatile("ds_read2_b32 %0, %1 offset0:1 offset1:17\n s_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(addr) : "memory");
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        u4 q;
        asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(addr) : "memory");
        b0 = q.x; b1 = q.y; b2 = q.z; b3 = q.w;
        if (k < n) { out[6*k] = (unsigned)a; out[6*k + 1] = (unsigned)(a >> 32); out[6*k + 2] = b0; out[6*k + 3] = b1; out[6*k + 4] = b2; out[6*k + 5] = b3; }
    }
}

__global__ void __launch_bounds__(64) probe(const unsigned char *src, unsigned *out32, unsigned long long *out64, int stride, int n)
{
    __shared__ __attribute__((aligned(16))) unsigned char buf[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) buf[i] = src[i];
    __syncthreads();
    for (int base = 0; base < n; base += 64)
    {
        const int k = base + (int)threadIdx.x;
        const unsigned addr = (unsigned)(size_t)(buf) + (unsigned)((k*stride) % 7900);     // every alignment, every bank, lane-dependent
        unsigned v; unsigned long long w;
        asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(addr) : "memory");
        if (k < n) { out32[k] = v; out64[k] = w; }
    }
}

int main()
{
    const int n = 1 << 14;
    std::vector<unsigned char> h(8192);
    unsigned seed = 777;
    for (auto &b : h) { seed = seed*1664525u + 1013904223u; b = (unsigned char)(seed >> 24); }
    unsigned char *dsrc; unsigned *d32; unsigned long long *d64;
    if (hipMalloc(&dsrc, 8192) != hipSuccess || hipMalloc(&d32, 4*n) != hipSuccess || hipMalloc(&d64, 8*n) != hipSuccess) { printf("no device\n"); return 0; }
    (void)hipMemcpy(dsrc, h.data(), 8192, hipMemcpyHostToDevice);
    const int strides[] = { 1, 3, 5, 17, 67, 68, 69, 129 };
    for (int s : strides)
    {
        std::vector<unsigned> o32(n); std::vector<unsigned long long> o64(n);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dsrc, d32, d64, s, n);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(o32.data(), d32, 4*n, hipMemcpyDeviceToHost);
        (void)hipMemcpy(o64.data(), d64, 8*n, hipMemcpyDeviceToHost);
        int bad32 = 0, first32 = -1, bad64 = 0, first64 = -1;
        for (int k = 0; k < n; k++)
        {
            const int a = (k*s) % 7900;
            unsigned w32 = 0; unsigned long long w64 = 0;
            for (int b = 0; b < 4; b++) w32 |= (unsigned)h[a + b] << (8*b);
            for (int b = 0; b < 8; b++) w64 |= (unsigned long long)h[a + b] << (8*b);
            if (w32 != o32[k]) { if (first32 < 0) first32 = k; bad32++; }
            if (w64 != o64[k]) { if (first64 < 0) first64 = k; bad64++; }
        }
        printf("b32_stride_%d mismatches=%d first=%d\n", s, bad32, first32);
        printf("b64_stride_%d mismatches=%d first=%d\n", s, bad64, first64);
        {
            unsigned *dw;
            std::vector<unsigned> ow(6*n);
            if (hipMalloc(&dw, 24*n) != hipSuccess) return 0;
            hipLaunchKernelGGL(probe_wide, dim3(1), dim3(64), 0, 0, dsrc, dw, s, n);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(ow.data(), dw, 24*n, hipMemcpyDeviceToHost);
            (void)hipFree(dw);
            int bad2 = 0, first2 = -1, bad128 = 0, first128 = -1;
            for (int k = 0; k < n; k++)
            {
                const int a = (k*s) % 7900;
                auto ld = [&](int o) { unsigned w = 0; for (int b = 0; b < 4; b++) w |= (unsigned)h[a + o + b] << (8*b); return w; };
                if (ow[6*k] != ld(4) || ow[6*k + 1] != ld(68)) { if (first2 < 0) first2 = k; bad2++; }
                if (ow[6*k + 2] != ld(0) || ow[6*k + 3] != ld(4) || ow[6*k + 4] != ld(8) || ow[6*k + 5] != ld(12)) { if (first128 < 0) first128 = k; bad128++; }
            }
            printf("read2_b32_stride_%d mismatches=%d first=%d\n", s, bad2, first2);
            printf("b128_stride_%d mismatches=%d first=%d\n", s, bad128, first128);
        }
    }
    return 0;
}
