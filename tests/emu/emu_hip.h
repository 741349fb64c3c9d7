/*
 * emu_hip.h -- TEST ONLY.  A host-memory stand-in for the handful of HIP runtime calls that h264-lab_amd/csrc/h264e_pool.h uses, so
 * that the product's host layer (pools, launch groups, submits, results) compiles unchanged into the emulation library and is tested
 * on a machine without a GPU: "device" memory is calloc'ed host memory, copies are memcpy, streams and events are no-ops because
 * every "launch" (emu_backend.cpp) runs to completion before it returns.
 */
#ifndef H264E_EMU_HIP_H
#define H264E_EMU_HIP_H
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <unistd.h>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorNotReady = 600, hipErrorOutOfMemory = 2 };
typedef struct emu_stream_tag { int unused; } *hipStream_t;
typedef struct emu_event_tag { int unused; } *hipEvent_t;
typedef void *hipDeviceptr_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
#define hipHostMallocMapped 1
#define hipHostMallocCoherent 2
#define hipHostMallocDefault 0

static inline const char *hipGetErrorString(hipError_t) { return "emulated HIP call failed"; }
static inline hipError_t hipGetLastError(void) { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
/* every emulation process is a device of its own: the per-device process guard never sees a second process */
static inline hipError_t hipDeviceGetPCIBusId(char *buf, int len, int) { snprintf(buf, (size_t)len, "emu_%ld", (long)getpid()); return hipSuccess; }
/* what the "device" can address: every hipMalloc / hipHostMalloc block is registered, so that the kernel sources' global-memory
 * accessors (wave.h cload / cstore / gload / dep_poll / g_atomic_*) can refuse an address that is NOT device memory -- e.g. a pointer
 * to a register or stack copy that was cast to a global pointer, which on the GPU is a memory fault (it was, once: DESIGN.md 8) */
void emu_register_device_block(void *p, size_t n);
void emu_unregister_device_block(void *p);
static inline hipError_t hipMalloc(void **p, size_t n) { *p = calloc(1, n ? n : 1); if (*p) emu_register_device_block(*p, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void *p) { if (p) emu_unregister_device_block(p); free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = calloc(1, n ? n : 1); if (*p) emu_register_device_block(*p, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void *p) { if (p) emu_unregister_device_block(p); free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t)
{
    for (size_t y = 0; y < h; y++) memcpy((char *)d + y*dp, (const char *)s + y*sp, w);
    return hipSuccess;
}
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetD32Async(hipDeviceptr_t d, int v, size_t count, hipStream_t) { for (size_t i = 0; i < count; i++) ((int *)d)[i] = v; return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = (hipStream_t)calloc(1, sizeof(**s)); return *s ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)calloc(1, sizeof(**e)); return *e ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0; return hipSuccess; }
#endif
