/*
 * wave.h -- the wave64 programming model the macroblock pipeline is written in.
 *
 * One workgroup = one 64-lane wavefront = one macroblock row.  The control flow of the
 * encoder is WAVE-UNIFORM: every lane executes the same scalar decisions on the same values
 * (registers replicated per lane, shared state in LDS), and pixel work fans out over the
 * lanes inside WAVE_FOR sections.  Cross-lane traffic goes through LDS (followed by
 * wave_sync()) or through the reductions below.
 *
 * Two builds of the same sources:
 *   - hipcc --offload-arch=gfx950 : the product (h264e_kernels.hip).
 *   - g++ -DH264E_EMU             : a lane-loop emulation used ONLY by tests/ to debug the
 *     kernel logic on a machine without a GPU.  It is never linked into libh264e_mi355x.so.
 */
#ifndef H264E_WAVE_H
#define H264E_WAVE_H

#include <stdint.h>
#include <string.h>

/* explicit address spaces: HBM pointers that were loaded from memory would otherwise be generic (FLAT instructions) */
#ifdef H264E_EMU
/* the emulation cannot type-check address spaces (GLOBAL_AS / LDS_AS are empty here), so it checks them at run time: every accessor of
 * memory that other workgroups or the host see asserts that the address lies in a block the "device" allocated (tests/emu/emu_backend.cpp) */
extern "C" void emu_check_global(const void *p, size_t n, const char *file, int line);
#define EMU_GLOBAL(p, n) emu_check_global((const void *)(p), (n), __FILE__, __LINE__)
#define GLOBAL_AS
#define LDS_AS
#define NOINLINE_DEV static __attribute__((noinline))
#define DEV static inline
#define DEVM inline                     /* member functions */
#define DCONST static const
#ifdef H264E_EMU_REVERSE      /* run lanes in the opposite order: catches code that leaks a lane-private value */
#define WAVE_FOR(l) for (int l = 63; l >= 0; --l)
#else
#define WAVE_FOR(l) for (int l = 0; l < 64; ++l)
#endif
struct u32x4 { uint32_t x, y, z, w; };
DEV void wave_sync() {}
DEV int wave_lane() { return 0; }
DEV int uni(int v) { return v; }
template <class F> DEV int wave_sum(F f)
{
    int s = 0;
    for (int l = 0; l < 64; ++l) s += f(l);
    return s;
}
/* four sums at once: f(lane, v[4]) */
template <class F> DEV void wave_sum4(F f, int out[4])
{
    out[0] = out[1] = out[2] = out[3] = 0;
    for (int l = 0; l < 64; ++l)
    {
        int v[4] = { 0, 0, 0, 0 };
        f(l, v);
        for (int k = 0; k < 4; ++k) out[k] += v[k];
    }
}
/* eight sums at once: f(lane, v[8]) */
template <class F> DEV void wave_sum8(F f, int out[8])
{
    for (int k = 0; k < 8; ++k) out[k] = 0;
    for (int l = 0; l < 64; ++l)
    {
        int v[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        f(l, v);
        for (int k = 0; k < 8; ++k) out[k] += v[k];
    }
}
template <class F> DEV uint64_t wave_ballot(F f)
{
    uint64_t m = 0;
    for (int l = 0; l < 64; ++l) if (f(l)) m |= 1ull << l;
    return m;
}
DEV uint32_t sad4_u8(uint32_t a, uint32_t b, uint32_t acc)
{
    for (int k = 0; k < 4; ++k)
    {
        int d = (int)((a >> (8*k)) & 255) - (int)((b >> (8*k)) & 255);
        acc += (uint32_t)(d < 0 ? -d : d);
    }
    return acc;
}
DEV int clz32(uint32_t v) { return __builtin_clz(v); }
DEV uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }

/* ---- lane groups (see the device build below): the emulation runs the four groups one after the other */
#define GRP_EACH(g) for (int g = 0; g < 4; ++g)
#ifdef H264E_EMU_REVERSE
#define GRP_FOR(i) for (int i = 15; i >= 0; --i)
#else
#define GRP_FOR(i) for (int i = 0; i < 16; ++i)
#endif
template <class F> DEV int grp_sum(F f)
{
    int s = 0;
    for (int i = 0; i < 16; ++i) s += f(i);
    return s;
}
template <class F> DEV void grp_sum4(F f, int out[4])
{
    out[0] = out[1] = out[2] = out[3] = 0;
    for (int i = 0; i < 16; ++i)
    {
        int v[4] = { 0, 0, 0, 0 };
        f(i, v);
        for (int k = 0; k < 4; ++k) out[k] += v[k];
    }
}
template <class F> DEV void grp_sum8(F f, int out[8])
{
    for (int k = 0; k < 8; ++k) out[k] = 0;
    for (int i = 0; i < 16; ++i)
    {
        int v[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        f(i, v);
        for (int k = 0; k < 8; ++k) out[k] += v[k];
    }
}
template <class F> DEV void grp_eval4(F f, int out[4]) { for (int d = 0; d < 4; ++d) out[d] = f(d); }
template <class F> DEV void grp_eval8(F f, int out[8]) { for (int d = 0; d < 8; ++d) out[d] = f(d); }
DEV void grp_count(int *p) { *p += 1; }

/* ---- V16 (see the device build below): the emulation keeps the 16 lanes of a tile in an array */
struct V16 { int v[16]; };
#define V16_BINOP(op) DEV V16 operator op(const V16 &a, const V16 &b) { V16 r; for (int i = 0; i < 16; i++) r.v[i] = a.v[i] op b.v[i]; return r; }
V16_BINOP(+) V16_BINOP(-) V16_BINOP(*) V16_BINOP(&) V16_BINOP(|)
#undef V16_BINOP
DEV V16 operator>>(const V16 &a, int s) { V16 r; for (int i = 0; i < 16; i++) r.v[i] = a.v[i] >> s; return r; }
DEV V16 v16_splat(int s) { V16 r; for (int i = 0; i < 16; i++) r.v[i] = s; return r; }
template <class F> DEV V16 v16_make(F f) { V16 r; for (int i = 0; i < 16; i++) r.v[i] = f(i, 0); return r; }            /* r[i] = f(i, tile) */
template <class F> DEV V16 v16_map(const V16 &a, F f) { V16 r; for (int i = 0; i < 16; i++) r.v[i] = f(i, a.v[i]); return r; }
template <class F> DEV V16 v16_map2(const V16 &a, const V16 &b, F f) { V16 r; for (int i = 0; i < 16; i++) r.v[i] = f(i, a.v[i], b.v[i]); return r; }
template <class F> DEV V16 v16_map3(const V16 &a, const V16 &b, const V16 &c, F f) { V16 r; for (int i = 0; i < 16; i++) r.v[i] = f(i, a.v[i], b.v[i], c.v[i]); return r; }
template <class F> DEV void v16_each(const V16 &a, F f) { for (int i = 0; i < 16; i++) f(i, 0, a.v[i]); }                  /* f(i, tile, value): stores */
/* does the wave hold a non-zero lane (uniform) / does MY tile hold one (per lane, 0 or 1) / which tiles do (bit t: tile t) */
DEV int v16_any(const V16 &a) { for (int i = 0; i < 16; i++) if (a.v[i]) return 1; return 0; }
DEV V16 v16_tile_any(const V16 &a) { V16 r; const int n = v16_any(a); for (int i = 0; i < 16; i++) r.v[i] = n; return r; }
DEV unsigned v16_tiles_nonzero(const V16 &a) { return (unsigned)v16_any(a); }
/* V16C (see the device build below): nothing to cache here; the context forms are the plain ones */
struct V16C { };
DEV V16C v16c_make() { return V16C(); }
template <class F> DEV V16 v16_make(const V16C &, F f) { return v16_make(f); }
template <class F> DEV V16 v16_map(const V16C &, const V16 &a, F f) { return v16_map(a, f); }
template <class F> DEV V16 v16_map2(const V16C &, const V16 &a, const V16 &b, F f) { return v16_map2(a, b, f); }
template <class F> DEV V16 v16_map3(const V16C &, const V16 &a, const V16 &b, const V16 &c, F f) { return v16_map3(a, b, c, f); }
template <class F> DEV void v16_each(const V16C &, const V16 &a, F f) { v16_each(a, f); }
DEV V16 v16_tile_any(const V16C &, const V16 &a) { return v16_tile_any(a); }
/* (the transforms themselves: enc_kernels.h v16_fwd4x4 / v16_inv4x4, found at the point of instantiation) */
template <class V> DEV V v16_fwd4x4_c(const V16C &, const V &d) { return v16_fwd4x4(d); }
template <class V> DEV V v16_inv4x4_c(const V16C &, const V &c) { return v16_inv4x4(c); }
template <int P0, int P1, int P2, int P3> DEV V16 v16_quadperm(const V16 &a)
{
    const int p[4] = { P0, P1, P2, P3 };
    V16 r;
    for (int i = 0; i < 16; i++) r.v[i] = a.v[(i & ~3) | p[i & 3]];
    return r;
}
DEV V16 v16_xpose(const V16 &a) { V16 r; for (int i = 0; i < 16; i++) r.v[i] = a.v[(i & 3)*4 + (i >> 2)]; return r; }
DEV V16 v16_from(const V16 &a, const V16 &idx) { V16 r; for (int i = 0; i < 16; i++) r.v[i] = a.v[idx.v[i] & 15]; return r; }   /* r[i] = a[idx[i]] */
/* bit i of the result: lane i's value is non-zero; tiles: how many of the wave's four tiles take part (the emulation runs one) */
DEV unsigned v16_nonzero_mask(const V16 &a) { unsigned m = 0; for (int i = 0; i < 16; i++) if (a.v[i]) m |= 1u << i; return m; }
#define V16_TILES 1

/* ---- small primitives with one definition per build (the kernel headers use these instead of switching on the build themselves) */
template <class P> DEV P uniptr(P p) { return p; }
DEV uint32_t alignbyte32(uint32_t hi, uint32_t lo, unsigned sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8*(sh & 3))); }
DEV uint32_t ld32_aligned(const void *q) { uint32_t v; memcpy(&v, q, 4); return v; }
DEV int opaque_int(int v) { return v; }
DEV int mul24(int a, int b) { return a*b; }           /* product of two values of at most 24 bits (the device build has a full-rate instruction for it) */
DEV int popc32(uint32_t v) { return __builtin_popcount(v); }
/* a progress counter of another workgroup: the emulation runs producers to completion before their consumers, so it is always there */
DEV int dep_poll(const int *p) { EMU_GLOBAL(p, 4); return 0x7fffffff; }
DEV void wave_nap() {}
DEV void consumer_acquire() {}
DEV void drain_stores() {}
DEV void g_atomic_add(int *p, int v) { EMU_GLOBAL(p, 4); *p += v; }
DEV int g_atomic_load(const int *p) { EMU_GLOBAL(p, 4); return *p; }
DEV void g_atomic_store(int *p, int v) { EMU_GLOBAL(p, 4); *p = v; }
/* V64: one value per lane of the wavefront that lives ACROSS lane sections (a register on the device; the emulation keeps the array) */
struct V64 { int v[64]; };
template <class F> DEV V64 v64_make(F f) { V64 r; WAVE_FOR(l) r.v[l] = f(l); return r; }
template <class F> DEV V64 v64_map(const V64 &a, F f) { V64 r; WAVE_FOR(l) r.v[l] = f(l, a.v[l]); return r; }
template <class F> DEV void v64_each(const V64 &a, F f) { WAVE_FOR(l) f(l, a.v[l]); }
DEV V64 v64_quad_sum(const V64 &a) { V64 r; for (int l = 0; l < 64; l++) r.v[l] = a.v[l & ~3] + a.v[(l & ~3) + 1] + a.v[(l & ~3) + 2] + a.v[(l & ~3) + 3]; return r; }
DEV int v64_read(const V64 &a, int lane) { return a.v[lane]; }
DEV int v64_own(const V64 &a, int lane) { return a.v[lane]; }          /* inside a lane section: this lane's own value */
template <class F> DEV void v64_each3(const V64 &a, const V64 &b, const V64 &c, F f) { WAVE_FOR(l) f(l, a.v[l], b.v[l], c.v[l]); }
/* the value of the lane below / above inside the 16-lane row; the row's first / last lane keeps its own */
DEV V64 v64_row_shr1(const V64 &a) { V64 r; for (int l = 0; l < 64; l++) r.v[l] = (l & 15) ? a.v[l - 1] : a.v[l]; return r; }
DEV V64 v64_row_shl1(const V64 &a) { V64 r; for (int l = 0; l < 64; l++) r.v[l] = (l & 15) != 15 ? a.v[l + 1] : a.v[l]; return r; }
/* minimum over the four quads of the 16-lane row, position by position inside the quad */
DEV V64 v64_row_quadmin(const V64 &a)
{
    V64 r;
    for (int l = 0; l < 64; l++) { int m = a.v[l]; for (int q = 0; q < 4; q++) { const int v = a.v[(l & 48) | (4*q) | (l & 3)]; if (v < m) m = v; } r.v[l] = m; }
    return r;
}
DEV uint64_t v64_nonzero_ballot(const V64 &a) { uint64_t m = 0; for (int l = 0; l < 64; l++) if (a.v[l]) m |= 1ull << l; return m; }
#define PROF_ROW_BEGIN(L) do { } while (0)
#define PROF_ROW_SYNC(L) do { } while (0)
#define PROF_ROW_END(L, C) do { } while (0)

#else /* device build */

#include <hip/hip_runtime.h>
#define EMU_GLOBAL(p, n) do { } while (0)
/* lane of the wavefront: a workgroup is ONE wavefront (64 threads) or, in the two-wave pipeline, two wavefronts with different jobs */
/* OPAQUE to the optimiser on purpose: with a plain expression the compiler computes every lane-dependent LDS address of the macroblock
 * loop once, in front of the loop, and keeps the lot alive in registers across the whole loop body (measured: two-wave kernel 168 VGPRs +
 * 61 spilled -> 164 and none; at 128 VGPRs 114 spilled -> 30; SGPR spills 289 -> 178).  Recomputing an address costs an instruction or two. */
static __device__ __forceinline__ int imin_raw(int a, int b) { return a < b ? a : b; }
static __device__ __forceinline__ int lane_opaque() { int l = (int)(threadIdx.x & 63u); asm volatile("" : "+v"(l)); return l; }
#define LANE lane_opaque()
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))
#define NOINLINE_DEV static __device__ __noinline__
#define DEV static __device__ __forceinline__
#define DEVM __device__ __forceinline__
#define DCONST static __device__ const
#define WAVE_FOR(l) for (int l = LANE, _w1 = 1; _w1; _w1 = 0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));
/* Single-wave workgroup: LDS operations of one wavefront execute in program order, so making one lane's LDS
 * store visible to another lane only needs the COMPILER to keep the order: a wavefront-scope fence (no
 * instruction) plus a wave barrier (keeps the lanes converged across it). */
DEV void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
DEV int wave_lane() { return LANE; }
/* a value every lane holds identically: move it to a scalar register so the control code runs on the scalar unit */
DEV int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
/* sum over the 64 lanes: DPP inside rows of 16 (quad swaps, half-row and row mirror), then 4 v_readlane */
DEV int wave_reduce_add(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);      /* quad_perm [1,0,3,2] */
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);      /* quad_perm [2,3,0,1] */
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);     /* row_half_mirror */
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);     /* row_mirror */
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) +
           __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
template <class F> DEV int wave_sum(F f) { return wave_reduce_add(f(LANE)); }
template <class F> DEV void wave_sum4(F f, int out[4])
{
    int v[4] = { 0, 0, 0, 0 };
    f(LANE, v);
    /* partial sums stay below 2^16 for every caller (<= 64 lanes x 4 x 255): reduce two per register */
    int a = v[0] | (v[1] << 16), b = v[2] | (v[3] << 16);
    a = wave_reduce_add(a);
    b = wave_reduce_add(b);
    out[0] = a & 0xffff; out[1] = (int)((unsigned)a >> 16);
    out[2] = b & 0xffff; out[3] = (int)((unsigned)b >> 16);
}
template <class F> DEV void wave_sum8(F f, int out[8])
{
    int v[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    f(LANE, v);
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
        int a = wave_reduce_add(v[2*k] | (v[2*k + 1] << 16));      /* sums stay below 2^16 for every caller */
        out[2*k] = a & 0xffff; out[2*k + 1] = (int)((unsigned)a >> 16);
    }
}
template <class F> DEV uint64_t wave_ballot(F f) { return __ballot(f(LANE)); }
DEV uint32_t sad4_u8(uint32_t a, uint32_t b, uint32_t acc) { return __builtin_amdgcn_sad_u8(a, b, acc); }
DEV int clz32(uint32_t v) { return __clz((int)v); }
DEV uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }

/*
 * Lane groups: the wavefront as FOUR independent 16-lane groups (one DPP row each).  Inside GRP_EACH(g) { ... } every group runs
 * its own control flow on its own values (plain per-lane registers that happen to agree inside a group; the hardware executes the
 * groups' paths under the exec mask) -- the motion search runs the four partition types side by side this way, one group each,
 * instead of one after the other.  Pixel work of a group fans out over its 16 lanes in GRP_FOR sections / the grp_sum reductions
 * (DPP inside the row: every lane of the group receives the sum); nothing in a group section may use the wave-level primitives
 * above (uni, wave_sum, LaneArr: they would mix the groups).
 */
#define GRP_EACH(g) for (int g = LANE >> 4, _e1 = 1; _e1; _e1 = 0)
#define GRP_FOR(i) for (int i = LANE & 15, _g1 = 1; _g1; _g1 = 0)
DEV int row_reduce_add(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);      /* quad_perm [1,0,3,2] */
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);      /* quad_perm [2,3,0,1] */
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);     /* row_half_mirror */
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);     /* row_mirror */
    return v;
}
template <class F> DEV int grp_sum(F f) { return row_reduce_add(f(LANE & 15)); }
template <class F> DEV void grp_sum4(F f, int out[4])
{
    int v[4] = { 0, 0, 0, 0 };
    f(LANE & 15, v);
    /* sums stay below 2^16 for every caller (<= 256 samples x 255): two per register */
    const int a = row_reduce_add(v[0] | (v[1] << 16)), b = row_reduce_add(v[2] | (v[3] << 16));
    out[0] = a & 0xffff; out[1] = (int)((unsigned)a >> 16);
    out[2] = b & 0xffff; out[3] = (int)((unsigned)b >> 16);
}
template <class F> DEV void grp_sum8(F f, int out[8])
{
    int v[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    f(LANE & 15, v);
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
        const int a = row_reduce_add(v[2*k] | (v[2*k + 1] << 16));
        out[2*k] = a & 0xffff; out[2*k + 1] = (int)((unsigned)a >> 16);
    }
}
/* out[d] = f(d) for d = 0..3 (0..7): a group-uniform function evaluated once per lane (lane & 3 picks d) and handed round inside the
 * quads, instead of four (eight) times in every lane */
template <class F> DEV void grp_eval4(F f, int out[4])
{
    const int c = f(LANE & 3);
    out[0] = __builtin_amdgcn_update_dpp(0, c, 0x00, 0xf, 0xf, true); out[1] = __builtin_amdgcn_update_dpp(0, c, 0x55, 0xf, 0xf, true);
    out[2] = __builtin_amdgcn_update_dpp(0, c, 0xAA, 0xf, 0xf, true); out[3] = __builtin_amdgcn_update_dpp(0, c, 0xFF, 0xf, 0xf, true);
}
template <class F> DEV void grp_eval8(F f, int out[8])
{
    const int c0 = f(LANE & 3), c1 = f(4 + (LANE & 3));
    out[0] = __builtin_amdgcn_update_dpp(0, c0, 0x00, 0xf, 0xf, true); out[1] = __builtin_amdgcn_update_dpp(0, c0, 0x55, 0xf, 0xf, true);
    out[2] = __builtin_amdgcn_update_dpp(0, c0, 0xAA, 0xf, 0xf, true); out[3] = __builtin_amdgcn_update_dpp(0, c0, 0xFF, 0xf, 0xf, true);
    out[4] = __builtin_amdgcn_update_dpp(0, c1, 0x00, 0xf, 0xf, true); out[5] = __builtin_amdgcn_update_dpp(0, c1, 0x55, 0xf, 0xf, true);
    out[6] = __builtin_amdgcn_update_dpp(0, c1, 0xAA, 0xf, 0xf, true); out[7] = __builtin_amdgcn_update_dpp(0, c1, 0xFF, 0xf, 0xf, true);
}
/* a statistics counter in LDS, bumped once per group */
DEV void grp_count(int *p) { if ((LANE & 15) == 0) __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

/*
 * V16: one value per lane of a 16-lane tile (a 4x4 block: lane i = 4*row + column, or coefficient index), for the 4x4 transforms,
 * the quantiser and the intra 4x4 reconstruction -- the data stays in registers and moves between lanes with DPP (inside the
 * quads) and ds_bpermute (transpose) instead of going through LDS between every step.  The wavefront holds FOUR tiles (one per DPP
 * row): kernels that work on one block (intra 4x4: a serial chain) run the same block in all four, kernels with independent blocks
 * (mb_write) give each row its own (tile index = LANE >> 4).  Written once for both builds: the emulation (above) keeps arrays.
 */
struct V16 { int v; };
#define V16_BINOP(op) DEV V16 operator op(const V16 &a, const V16 &b) { V16 r; r.v = a.v op b.v; return r; }
V16_BINOP(+) V16_BINOP(-) V16_BINOP(*) V16_BINOP(&) V16_BINOP(|)
#undef V16_BINOP
DEV V16 operator>>(const V16 &a, int s) { V16 r; r.v = a.v >> s; return r; }
DEV V16 v16_splat(int s) { V16 r; r.v = s; return r; }
template <class F> DEV V16 v16_make(F f) { V16 r; r.v = f(LANE & 15, LANE >> 4); return r; }
template <class F> DEV V16 v16_map(const V16 &a, F f) { V16 r; r.v = f(LANE & 15, a.v); return r; }
template <class F> DEV V16 v16_map2(const V16 &a, const V16 &b, F f) { V16 r; r.v = f(LANE & 15, a.v, b.v); return r; }
template <class F> DEV V16 v16_map3(const V16 &a, const V16 &b, const V16 &c, F f) { V16 r; r.v = f(LANE & 15, a.v, b.v, c.v); return r; }
template <class F> DEV void v16_each(const V16 &a, F f) { f(LANE & 15, LANE >> 4, a.v); }
/* does the wave hold a non-zero lane (uniform) / does MY tile hold one (per lane, 0 or 1) / which tiles do (bit t: tile t): one ballot each */
DEV int v16_any(const V16 &a) { return __ballot(a.v != 0) != 0; }
DEV V16 v16_tile_any(const V16 &a)
{
    const unsigned long long m = __ballot(a.v != 0);
    const int t = LANE >> 4;
    const unsigned w = (t & 2) ? (unsigned)(m >> 32) : (unsigned)m;
    V16 r;
    r.v = ((w >> (16*(t & 1))) & 0xffffu) != 0;
    return r;
}
DEV unsigned v16_tiles_nonzero(const V16 &a)
{
    const unsigned long long m = __ballot(a.v != 0);
    const unsigned lo = (unsigned)m, hi = (unsigned)(m >> 32);
    return ((lo & 0xffffu) ? 1u : 0u) | ((lo >> 16) ? 2u : 0u) | ((hi & 0xffffu) ? 4u : 0u) | ((hi >> 16) ? 8u : 0u);
}
template <int P0, int P1, int P2, int P3> DEV V16 v16_quadperm(const V16 &a)
{
    V16 r;
    r.v = __builtin_amdgcn_update_dpp(0, a.v, P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xf, 0xf, true);
    return r;
}
DEV V16 v16_xpose(const V16 &a)
{
    V16 r;
    r.v = __builtin_amdgcn_ds_bpermute(4*((LANE & 48) | ((LANE & 3) << 2) | ((LANE >> 2) & 3)), a.v);
    return r;
}
DEV V16 v16_from(const V16 &a, const V16 &idx)
{
    V16 r;
    r.v = __builtin_amdgcn_ds_bpermute(4*((LANE & 48) | (idx.v & 15)), a.v);
    return r;
}
/*
 * V16C: what a lane needs again and again inside one V16 kernel -- its position in the tile, and the per-lane constants of the 4x4
 * transforms -- computed ONCE from the (opaque) lane id.  The plain forms above read the lane id anew in every call, which is what keeps
 * the compiler from hoisting lane-dependent values out of the macroblock loop (lane_opaque) -- and costs a dozen instructions per call;
 * a kernel that runs many V16 steps back to back (mb_write's fused transform pass, the intra 4x4 block coder) makes one context and
 * hands it to the context forms below: their lane-dependent values are ordinary values the compiler shares inside that kernel.
 */
struct V16C
{
    int i, tile;                /* lane & 15, lane >> 4 */
    int xp;                     /* ds_bpermute address of the transposed position inside the tile */
    int f_s, f_ka, f_kb;        /* forward pass: u = partner + f_s*own; out = f_ka*A + f_kb*B */
    int i_sh, i_a1, i_a2, i_s;  /* inverse pass: u = i_a1*(own >> i_sh) + i_a2*partner; out = A + i_s*B */
};
DEV V16C v16c_make()
{
    V16C c;
    const int l = LANE, x = l & 3;
    c.i = l & 15; c.tile = l >> 4;
    c.xp = 4*((l & 48) | ((l & 3) << 2) | ((l >> 2) & 3));
    c.f_s = (x & 2) ? -1 : 1;
    c.f_ka = x == 1 ? 2 : 1; c.f_kb = x == 0 ? 1 : x == 1 ? 1 : x == 2 ? -1 : -2;
    c.i_sh = x & 1; c.i_a1 = x == 2 ? -1 : 1; c.i_a2 = x == 1 ? -1 : 1; c.i_s = (x & 2) ? -1 : 1;
    return c;
}
template <class F> DEV V16 v16_make(const V16C &c, F f) { V16 r; r.v = f(c.i, c.tile); return r; }
template <class F> DEV V16 v16_map(const V16C &c, const V16 &a, F f) { V16 r; r.v = f(c.i, a.v); return r; }
template <class F> DEV V16 v16_map2(const V16C &c, const V16 &a, const V16 &b, F f) { V16 r; r.v = f(c.i, a.v, b.v); return r; }
template <class F> DEV V16 v16_map3(const V16C &c, const V16 &a, const V16 &b, const V16 &d, F f) { V16 r; r.v = f(c.i, a.v, b.v, d.v); return r; }
template <class F> DEV void v16_each(const V16C &c, const V16 &a, F f) { f(c.i, c.tile, a.v); }
DEV V16 v16_tile_any(const V16C &c, const V16 &a)
{
    const unsigned long long m = __ballot(a.v != 0);
    const unsigned w = (c.tile & 2) ? (unsigned)(m >> 32) : (unsigned)m;
    V16 r;
    r.v = ((w >> (16*(c.tile & 1))) & 0xffffu) != 0;
    return r;
}
/* the 4x4 core transforms with the lane constants of the context: every step is a DPP move inside the quad and a 24-bit multiply-add
 * with a per-lane coefficient (the operands are 16-bit quantities: v_mul_i32_i24 is exact and full rate), the other direction goes
 * through one ds_bpermute transpose.  Same arithmetic as enc_kernels.h v16_fwd4x4 / v16_inv4x4, which state it in the open (the
 * emulation build runs those; tests/test_stages.py holds both against the reference's own functions). */
#define V16_DPP(v, p0, p1, p2, p3) __builtin_amdgcn_update_dpp(0, (v), (p0) | ((p1) << 2) | ((p2) << 4) | ((p3) << 6), 0xf, 0xf, true)
DEV int v16_fwd_quad_c(const V16C &c, int d)
{
    const int u = V16_DPP(d, 3, 2, 1, 0) + __mul24(d, c.f_s);                    /* t0 = d0+d3, t2 = d1+d2, t3 = d1-d2, t1 = d0-d3 */
    return __mul24(V16_DPP(u, 0, 3, 0, 3), c.f_ka) + __mul24(V16_DPP(u, 1, 2, 1, 2), c.f_kb);
}
DEV int v16_inv_quad_c(const V16C &c, int d)
{
    const int u = __mul24(d >> c.i_sh, c.i_a1) + __mul24(V16_DPP(d, 2, 3, 0, 1), c.i_a2);     /* e0 e2 e1 e3 */
    return V16_DPP(u, 0, 2, 2, 0) + __mul24(V16_DPP(u, 3, 1, 1, 3), c.i_s);
}
template <class V> DEV V v16_fwd4x4_c(const V16C &c, const V &d)
{
    V r;
    r.v = v16_fwd_quad_c(c, __builtin_amdgcn_ds_bpermute(c.xp, v16_fwd_quad_c(c, d.v)));
    return r;
}
template <class V> DEV V v16_inv4x4_c(const V16C &c, const V &q)
{
    const int f = (int16_t)v16_inv_quad_c(c, __builtin_amdgcn_ds_bpermute(c.xp, q.v));             /* int16 between the passes, H:2436-2489 */
    const int g = v16_inv_quad_c(c, __builtin_amdgcn_ds_bpermute(c.xp, f));
    V r;
    r.v = (int16_t)((g + 32) >> 6);
    return r;
}
#undef V16_DPP
/* bits 16t .. 16t+15 of the ballot belong to tile t */
DEV unsigned long long v16_nonzero_ballot(const V16 &a) { return __ballot(a.v != 0); }
DEV unsigned v16_nonzero_mask(const V16 &a) { return (unsigned)(__ballot(a.v != 0) & 0xffffu); }
#define V16_TILES 4

/* ---- small primitives with one definition per build (the kernel headers use these instead of switching on the build themselves) */
template <class P> DEV P uniptr(P p)
{
    const unsigned long long v = (unsigned long long)(uintptr_t)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (P)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}
DEV uint32_t alignbyte32(uint32_t hi, uint32_t lo, unsigned sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
DEV uint32_t ld32_aligned(const LDS_AS uint32_t *q) { return *q; }
/* keeps the compiler from looking through a value (enc_kernels.h shr_opaque: the v_ashr_pk_u8_i32 finding) */
DEV int opaque_int(int v) { asm volatile("" : "+v"(v)); return v; }
/* low 32 bits of the product of two values of at most 24 bits: v_mul_i32_i24, full rate (a 32-bit v_mul_lo / v_mad_u64 runs at a quarter) */
DEV int mul24(int a, int b) { return __mul24(a, b); }
DEV int popc32(uint32_t v) { return __popc(v); }
/* a progress counter of another workgroup, read past the caches' stale copies (relaxed, agent scope: an sc1 load) */
DEV int dep_poll(const GLOBAL_AS int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV void wave_nap() { __builtin_amdgcn_s_sleep(8); }
DEV void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
DEV void g_atomic_add(GLOBAL_AS int *p, int v) { if (v) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV int g_atomic_load(const GLOBAL_AS int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV void g_atomic_store(GLOBAL_AS int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
/* V64: one value per lane of the wavefront that lives ACROSS lane sections: a register */
struct V64 { int v; };
template <class F> DEV V64 v64_make(F f) { V64 r; r.v = f(LANE); return r; }
template <class F> DEV V64 v64_map(const V64 &a, F f) { V64 r; r.v = f(LANE, a.v); return r; }
template <class F> DEV void v64_each(const V64 &a, F f) { f(LANE, a.v); }
DEV V64 v64_quad_sum(const V64 &a)
{
    V64 r;
    r.v = a.v + __builtin_amdgcn_update_dpp(0, a.v, 0xB1, 0xf, 0xf, true);      /* quad_perm [1,0,3,2] */
    r.v += __builtin_amdgcn_update_dpp(0, r.v, 0x4E, 0xf, 0xf, true);           /* quad_perm [2,3,0,1] */
    return r;
}
DEV int v64_read(const V64 &a, int lane) { return __builtin_amdgcn_readlane(a.v, lane); }
DEV int v64_own(const V64 &a, int) { return a.v; }                     /* inside a lane section: this lane's own value */
template <class F> DEV void v64_each3(const V64 &a, const V64 &b, const V64 &c, F f) { f(LANE, a.v, b.v, c.v); }
/* the value of the lane below / above inside the 16-lane row (DPP row_shr:1 / row_shl:1); the row's first / last lane keeps its own */
DEV V64 v64_row_shr1(const V64 &a) { V64 r; r.v = __builtin_amdgcn_update_dpp(a.v, a.v, 0x111, 0xf, 0xf, false); return r; }
DEV V64 v64_row_shl1(const V64 &a) { V64 r; r.v = __builtin_amdgcn_update_dpp(a.v, a.v, 0x101, 0xf, 0xf, false); return r; }
/* minimum over the four quads of the 16-lane row, position by position inside the quad (DPP row_ror:4, row_ror:8) */
DEV V64 v64_row_quadmin(const V64 &a)
{
    V64 r;
    r.v = imin_raw(a.v, __builtin_amdgcn_update_dpp(a.v, a.v, 0x124, 0xf, 0xf, false));
    r.v = imin_raw(r.v, __builtin_amdgcn_update_dpp(r.v, r.v, 0x128, 0xf, 0xf, false));
    return r;
}
DEV uint64_t v64_nonzero_ballot(const V64 &a) { return __ballot(a.v != 0); }
#endif

/* diagnostic build only (-DH264E_STAMPS): cycle stamps per pipeline phase, accumulated in LDS (never in the product) */
#if defined(H264E_STAMPS) && !defined(H264E_EMU)
#define PROF_W ((int)(threadIdx.x >> 6))         /* every wavefront of the workgroup keeps its own stamps */
#define STAMP(L, id) do { unsigned long long t_ = __builtin_readcyclecounter(); (L).prof[PROF_W][id] += t_ - (L).prof_last[PROF_W]; (L).prof_last[PROF_W] = t_; } while (0)
#define PCOUNT(L, id) do { (L).prof[PROF_W][id]++; } while (0)
#define PTIC() unsigned long long tic_ = __builtin_readcyclecounter()
#define PTOC(L, id) do { (L).prof[PROF_W][id] += __builtin_readcyclecounter() - tic_; } while (0)
#define PROF_ROW_BEGIN(L) do { (L).prof_c0 = __builtin_readcyclecounter(); (L).prof_w0 = wall_clock64(); } while (0)     /* shader-clock cycles vs constant 100 MHz clock: effective frequency */
#define PROF_ROW_SYNC(L) do { (L).prof_last[0] = (L).prof_last[1] = (L).prof_last[2] = (L).prof_last[3] = __builtin_readcyclecounter(); } while (0)
#define PROF_ROW_END(L, C) do { (L).prof[PROF_W][28] = __builtin_readcyclecounter() - (L).prof_c0; (L).prof[PROF_W][29] = wall_clock64() - (L).prof_w0; wave_sync(); \
                                if (LANE < 32 && (C).prof) atomicAdd((C).prof + LANE, (L).prof[PROF_W][LANE]); } while (0)
#else
#define STAMP(L, id) do { } while (0)
#define PCOUNT(L, id) do { } while (0)
#define PTIC() do { } while (0)
#define PTOC(L, id) do { } while (0)
#ifndef PROF_ROW_BEGIN
#define PROF_ROW_BEGIN(L) do { } while (0)
#define PROF_ROW_SYNC(L) do { } while (0)
#define PROF_ROW_END(L, C) do { } while (0)
#endif
#endif

/*
 * A small array of wave-uniform ints indexed by a wave-uniform index, held in ONE vector register (lane i = element i):
 * v_readlane / a lane-select write take a few cycles where an LDS-resident scalar costs a full LDS round trip per access.
 * Only for the uniform control code (every lane calls get/set with the same arguments).
 */
struct LaneArr
{
#ifdef H264E_EMU
    int a[64];
    void clear() { for (int i = 0; i < 64; i++) a[i] = 0; }
    int get(int i) const { return a[i]; }
    void set(int i, int v) { a[i] = v; }
    int has(int v, int n) const { for (int i = 0; i < n; i++) if (a[i] == v) return 1; return 0; }     /* is v among the first n elements? */
    int get_any(int i) const { return a[i & 63]; }       /* inside a lane-group section: the index may differ from group to group */
#else
    int r;
    __device__ __forceinline__ void clear() { r = 0; }
    __device__ __forceinline__ int get(int i) const { return __builtin_amdgcn_readlane(r, __builtin_amdgcn_readfirstlane(i)); }
    __device__ __forceinline__ void set(int i, int v) { r = (LANE == __builtin_amdgcn_readfirstlane(i)) ? __builtin_amdgcn_readfirstlane(v) : r; }    /* compare + select: no v_writelane builtin */
    __device__ __forceinline__ int has(int v, int n) const { return __ballot(LANE < n && r == __builtin_amdgcn_readfirstlane(v)) != 0; }
    __device__ __forceinline__ int get_any(int i) const { return __builtin_amdgcn_ds_bpermute(4*(i & 63), r); }     /* per-lane index (lane groups) */
#endif
};

typedef LDS_AS uint8_t lu8;                           /* a byte in LDS (explicit, so out-of-line functions keep ds_* instructions) */
typedef GLOBAL_AS uint8_t gu8;                        /* a byte in HBM */
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
typedef GLOBAL_AS u32_unaligned gu32u;

DEV uint32_t gload32(const gu8 *p) { EMU_GLOBAL(p, 4); return *(const gu32u *)p; }
DEV void gstore32(gu8 *p, uint32_t v) { EMU_GLOBAL(p, 4); *(gu32u *)p = v; }

/*
 * Coherent accessors for every byte that one workgroup writes and another reads INSIDE a launch (pictures, neighbour
 * records, pending lines, bit buffers, macroblock records): `sc1` write-through stores and `sc1` loads (relaxed
 * agent-scope atomics).  With every such byte stored and loaded this way, a hand-off needs no agent-scope release
 * (L2 write-back) and no acquire (L1 invalidate) -- only: stores, s_waitcnt vmcnt(0), counter store | counter poll,
 * loads (MI355X_MICROARCH.md "Valid forms").  Naturally aligned 4- or 8-byte accesses only.
 */
#ifdef H264E_EMU
#define H264E_PLAIN_UNALIGNED_LOADS 0
DEV uint32_t cload32(const gu8 *p) { uint32_t v; EMU_GLOBAL(p, 4); memcpy(&v, p, 4); return v; }
DEV uint64_t cload64(const gu8 *p) { uint64_t v; EMU_GLOBAL(p, 8); memcpy(&v, p, 8); return v; }
DEV u32x4 cload128(const gu8 *p) { u32x4 v; EMU_GLOBAL(p, 16); memcpy(&v, p, 16); return v; }
DEV void cstore32(gu8 *p, uint32_t v) { EMU_GLOBAL(p, 4); memcpy(p, &v, 4); }
DEV void cstore64(gu8 *p, uint64_t v) { EMU_GLOBAL(p, 8); memcpy(p, &v, 8); }
#else
/* H264E_COHERENT_LOADS = 1: sc1 loads, the consumer needs no acquire.  0: plain loads behind ONE agent-scope acquire per
 * hand-off (invalidates the CU's L1), the form cdna_hip_programming.md Guideline 16 recommends with write-through stores. */
#ifndef H264E_COHERENT_LOADS
#define H264E_COHERENT_LOADS 0
#endif
/* plain (cached) loads may be unaligned dword loads; the coherent sc1 form needs natural alignment (enc_kernels.h ref_load4) */
#define H264E_PLAIN_UNALIGNED_LOADS (!H264E_COHERENT_LOADS)
#if H264E_COHERENT_LOADS
DEV uint32_t cload32(const gu8 *p) { return __hip_atomic_load((const GLOBAL_AS uint32_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV uint64_t cload64(const gu8 *p) { return __hip_atomic_load((const GLOBAL_AS uint64_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV void consumer_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
#else
DEV uint32_t cload32(const gu8 *p) { return *(const GLOBAL_AS uint32_t *)p; }
DEV uint64_t cload64(const gu8 *p) { return *(const GLOBAL_AS uint64_t *)p; }
DEV u32x4 cload128(const gu8 *p) { return *(const GLOBAL_AS u32x4 *)p; }           /* 16 bytes, dword aligned (plain, behind the hand-off's acquire) */
DEV void consumer_acquire()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
#endif
DEV void cstore32(gu8 *p, uint32_t v) { __hip_atomic_store((GLOBAL_AS uint32_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV void cstore64(gu8 *p, uint64_t v) { __hip_atomic_store((GLOBAL_AS uint64_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif

DEV int imin(int a, int b) { return a < b ? a : b; }
DEV int imax(int a, int b) { return a > b ? a : b; }
DEV int iabs(int x) { return x < 0 ? -x : x; }
DEV int clip255(int x) { return x < 0 ? 0 : x > 255 ? 255 : x; }
DEV int clip3(int lo, int hi, int x) { return x < lo ? lo : x > hi ? hi : x; }

#endif
