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

// hardware fp32 atomic add without return (global_atomic_add_f32), device scope
__device__ __forceinline__ void atomic_add_f32(float *p, float v) { unsafeAtomicAdd(p, v); }

}  // namespace lsn
