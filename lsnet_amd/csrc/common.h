// Shared helpers for the gfx950 kernels of liblsnet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/lsnet_hip.h"

namespace lsn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// thread-local error text returned by lsn_last_error()
char *err_buf();
int fail(int code, const char *fmt, ...);

#define LSN_CHECK(cond, ...)                                  \
    do {                                                      \
        if (!(cond)) return lsn::fail(LSN_ERR_INVALID, __VA_ARGS__); \
    } while (0)

#define LSN_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return lsn::fail(LSN_ERR_RUNTIME, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                             __FILE__, __LINE__);                                             \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// lsn_scratch_stats counters (conv.hip): what the library itself asks of the HIP runtime outside launches -- a training loop
// whose shapes change every iteration (multi-scale training) must reach a state in which none of these moves any more
enum { STAT_MALLOCS = 0, STAT_HELD_BYTES = 1, STAT_BLOCKING_SYNCS = 2, STAT_POOL_ALLOCS = 3 };
void lib_stat(int which, long long add);

// v_mfma_f32_32x32x2_f32: D(32x32) += A(32x2) * B(2x32).
//   lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31];
//   D register r of lane l is D[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// v_mfma_f32_16x16x4_f32: D(16x16) += A(16x4) * B(4x16).
//   lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15];
//   D register r of lane l is D[row = 4 * (l >> 4) + r][col = l & 15].
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---- split-bf16 ("bf16x3") products on the matrix pipe ------------------------------------------------
// fp32 MFMA runs on the same ALUs as fp32 VALU work at 1/16 of the bf16 MFMA rate (MI355X_MICROARCH.md).  A
// product a*b of fp32 numbers is instead computed as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with a = a_hi + a_lo
// split into two bf16 values (8 + 8 mantissa bits) and fp32 accumulation: relative error <= 2^-16 per product
// (the dropped a_lo*b_lo term and the 2^-17 truncation of each operand), on v_mfma_f32_32x32x16_bf16.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// A: lane l holds row (l & 31), k = 8 (l >> 5) .. + 7;  B: column (l & 31), same k;  D as the fp32 32x32 form
__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// A: lane l holds row (l & 15), k = 8 (l >> 4) .. + 7;  B: column (l & 15), same k;  D as the fp32 16x16 form
__device__ __forceinline__ f32x4 mfma16_bf16(bf16x8 a, bf16x8 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// bf16(v0) | bf16(v1) << 16, round to nearest even: ONE v_cvt_pk_bf16_f32.  (Written as asm because hipcc, given the
// equivalent vector conversion, also converts element 0 on its own and re-packs with v_bfi: 16 instead of 11 VALU
// operations per 3-way split of a pair, in kernels whose staging instruction stream is what limits them.)
__device__ __forceinline__ unsigned cvt_pk_bf16(float v0, float v1)
{
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(v0), "v"(v1));
    return r;
}

// (v0, v1) -> packed bf16 pairs hi = bf16(v), lo = bf16(v - hi); element 0 in the low half-word
__device__ __forceinline__ void split_bf16x2(float v0, float v1, unsigned &hi, unsigned &lo)
{
    hi = cvt_pk_bf16(v0, v1);
    const float h0 = __uint_as_float(hi << 16), h1 = __uint_as_float(hi & 0xffff0000u);
    lo = cvt_pk_bf16(v0 - h0, v1 - h1);
}

// ---- fp32-equivalent products on the matrix pipe ("bf16x6") -------------------------------------------
// a = a_h + a_m + a_l with three bf16 values (8 + 8 + 8 mantissa bits: the split is EXACT for every fp32 number
// whose exponent leaves room for the low part, i.e. all but subnormal-scale values) and
//   a*b = h*h + h*m + m*h + m*m + h*l + l*h          (dropped: m*l, l*m, l*l <= 2^-25 relative)
// six bf16 MFMAs with fp32 accumulation: the per-product error is below fp32's own product rounding (2^-24),
// so a contraction differs from an fmaf chain only by summation order.  Peak: 2516 / 6 = 419 TFLOP/s of
// algorithmic flops, 2.7x the fp32 MFMA pipe.
__device__ __forceinline__ void split3_bf16x2(float v0, float v1, unsigned &hi, unsigned &mid, unsigned &lo)
{
    hi = cvt_pk_bf16(v0, v1);
    const float r0 = v0 - __uint_as_float(hi << 16), r1 = v1 - __uint_as_float(hi & 0xffff0000u);
    mid = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(mid << 16), s1 = r1 - __uint_as_float(mid & 0xffff0000u);
    lo = cvt_pk_bf16(s0, s1);
}

// NP = number of bf16 products per fp32 product (3: "bf16x3", 6: "bf16x6"); planes per operand and the
// (A plane, B plane) of product q.  Small terms are issued last so that a consumer may stop early.
template <int NP> struct SplitCfg;
template <> struct SplitCfg<3> {
    static constexpr int NPL = 2;
    __host__ __device__ static constexpr int pa(int q) { return q == 2 ? 1 : 0; }
    __host__ __device__ static constexpr int pb(int q) { return q == 1 ? 1 : 0; }
};
template <> struct SplitCfg<6> {
    static constexpr int NPL = 3;
    //            q:  0      1      2      3      4      5
    //            A:  h      h      m      m      h      l
    //            B:  h      m      h      m      l      h
    __host__ __device__ static constexpr int pa(int q) { return q == 0 ? 0 : q == 1 ? 0 : q == 2 ? 1 : q == 3 ? 1 : q == 4 ? 0 : 2; }
    __host__ __device__ static constexpr int pb(int q) { return q == 0 ? 0 : q == 1 ? 1 : q == 2 ? 0 : q == 3 ? 1 : q == 4 ? 2 : 0; }
};

// (v0, v1) -> NPL packed bf16 pairs
template <int NPL>
__device__ __forceinline__ void split_planes(float v0, float v1, unsigned (&p)[NPL])
{
    if constexpr (NPL == 2)
        split_bf16x2(v0, v1, p[0], p[1]);
    else
        split3_bf16x2(v0, v1, p[0], p[1], p[2]);
}

// XCD-aware work order.  Workgroups are dealt round-robin to the 8 XCDs (linear id b -> XCD b % 8), each with its
// own 4 MB L2.  Giving XCD x the CONTIGUOUS range of work items [start(x), start(x+1)) instead of every 8th one
// keeps spatially adjacent tiles -- which sample the same input rows and the same gout rows -- behind one L2
// (measured before: the forward kernel fetched 443 MB over the fabric for a 46 MB input).  Bijection on [0, total).
__device__ __forceinline__ int xcd_remap(int b, int total)
{
    const int q = total >> 3, r = total & 7, x = b & 7, i = b >> 3;
    return x * q + min(x, r) + i;
}

// ---- handing data to ANOTHER workgroup of the same launch (last-arriver reductions) ----
// The eight XCDs have their own L2s.  A release / acquire fence pair (__threadfence) makes a workgroup's plain stores visible
// to the others by writing back and invalidating the WHOLE L2 -- measured at 150 - 250 us per launch when a few hundred
// workgroups do it (profiles/r5_sk_fence.txt; the first ticketed GroupNorm statistics lost 0.4 ms per step to it).  Instead
// the handed-over values are written and read with AGENT-scope accesses (sc1: through to / from the coherence point), the
// writer waits for its stores (vmcnt) before it draws the ticket, and nothing is flushed.
template <typename T> __device__ __forceinline__ void store_agent(T *p, T v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T> __device__ __forceinline__ T load_agent(const T *p)
{
    return __hip_atomic_load(const_cast<T *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wait_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// hardware fp32 atomic add without return (global_atomic_add_f32), device scope
__device__ __forceinline__ void atomic_add_f32(float *p, float v) { unsafeAtomicAdd(p, v); }

}  // namespace lsn
