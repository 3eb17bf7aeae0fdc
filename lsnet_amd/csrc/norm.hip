// GroupNorm (+ optional ReLU) forward / backward on channels-last fp32 tensors, several tensors (FPN levels
// sharing one GroupNorm module) per launch.
//
// The reference calls torch.nn.GroupNorm + nn.ReLU (lsnet_head.py:1830-1849, 136-141; fpn.py:65-156 through
// ConvModule): three ATen kernels forward, five backward, on NCHW tensors.  On MI355X the hot path keeps
// activations in NHWC for the deformable-conv gathers; ATen's GroupNorm returns NCHW-contiguous tensors, which
// cost a layout copy before and after every deformable conv (324 permute launches per training step).  These
// kernels read and write NHWC directly and fold the ReLU in:
//   forward : gn_stats_kernel (shifted sums -> fp64 atomics per (image, group)), gn_apply_kernel (y = x*a + b)
//   backward: gn_bwd_reduce_kernel (per (image, channel) sums of dy and dy*xhat), gn_bwd_imgsum_kernel (the same per image,
//             d gamma / d beta by its last block), gn_bwd_apply_kernel
// All are HBM-bound streaming kernels: one pixel row (C floats) is read by C/4 lanes as float4.
// Numerics: variance from shifted sums (shift = the group's first element of the image), accumulated in fp32
// per thread and in fp64 across threads / blocks -- robust when |mean| >> std.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lsnet_hip.h"
#include "common.h"
#include "prof.h"

namespace lsn {

int lib_tickets(unsigned **p, hipStream_t st);   // conv.hip

constexpr int GN_MAXLV = 16;
constexpr int GN_PIX = 64;    // pixels per block

struct GnLvl {
    const float *x, *dy;
    float *y, *dx;
    int B, HW;
    long long ybs, dybs;   // floats between the images of y / dy (HW * C when dense)
    int tile0;   // first block of this level
    int img0;    // first (image) slot of this level in the statistics buffers
};
struct GnArgs {
    GnLvl lv[GN_MAXLV];
    int nlv, C, G, relu;
    float eps;
    const float *gamma, *beta;
    double *sums;      // [images][G][2]   shifted sum, shifted sum of squares   (zero when the statistics kernel starts)
    double *clear;     // the statistics kernel zeroes clear[0 .. nclear): the sums the PREVIOUS call of this stream left behind
    int nclear;
    unsigned *ticket;  // self-resetting counters of this stream (conv.hip lib_tickets); [1]: the backward's image sums
    float *mean_rstd;  // [images][G][2]
    float *ab;         // [images][C][2]   backward: sum dy*xhat, sum dy
    float *part;       // [blocks][C][2]   backward: the same per 64-pixel block (summed per image in a fixed order)
    float *dgamma, *dbeta;
};

__device__ __forceinline__ const GnLvl &gn_level(const GnArgs &a, int tile, int &li)
{
    li = 0;
    while (li + 1 < a.nlv && tile >= a.lv[li + 1].tile0) ++li;
    return a.lv[li];
}

// block -> (level, image, pixel range); thread -> (channel quad q, row slot)
struct GnPos {
    int b, p0, p1, q, row, rows;
};
__device__ __forceinline__ GnPos gn_pos(const GnArgs &a, const GnLvl &L)
{
    GnPos r;
    const int tpi = (L.HW + GN_PIX - 1) / GN_PIX;   // tiles per image
    const int t = blockIdx.x - L.tile0;
    r.b = t / tpi;
    r.p0 = (t - r.b * tpi) * GN_PIX;
    r.p1 = min(r.p0 + GN_PIX, L.HW);
    const int qn = a.C >> 2;
    r.q = threadIdx.x % qn;
    r.row = threadIdx.x / qn;
    r.rows = 256 / qn;
    return r;
}

// (Round 5 tried the statistics without the fp64 atomics: per-block partials + a ticket, the last block of
// an image adding them in block order -- correct and bit-stable, but 0.1 ms per step SLOWER than this form, 32.60 vs 32.50 ms,
// old and new library alternating on one box: the finisher's tail costs more than 16 fills and 45 k atomics.  With
// __threadfence() instead of agent-scope stores it lost 0.4 ms.  profiles/r5_gn_ticket.txt.  The fills went another way:
// gn_sums() below.)
// EG ("element groups"): fewer than four channels per group (C / G = 1 or 2: the test-size heads, 32 channels in 32 groups, 16 in
// 8) -- the four channels of a thread's quad belong to different groups, so shift, moments and group sums are kept per element.
template <bool EG>
__global__ __launch_bounds__(256) void gn_stats_kernel(const GnArgs a)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.nclear; i += gridDim.x * 256) a.clear[i] = 0.0;
    int li;
    const GnLvl &L = gn_level(a, blockIdx.x, li);
    const GnPos p = gn_pos(a, L);
    const int cpg = a.C / a.G, g = (p.q * 4) / cpg;
    const float *xb = L.x + (size_t)p.b * L.HW * a.C;
    __shared__ double acc[256 * 2 * (EG ? 4 : 1)];
    const int qn = a.C >> 2;
    if constexpr (EG) {
        float K[4], s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) K[e] = xb[(p.q * 4 + e) / cpg * cpg];   // shift: first element of the element's group
#pragma unroll 4
        for (int px = p.p0 + p.row; px < p.p1; px += p.rows) {
            const float4 v = *reinterpret_cast<const float4 *>(xb + (size_t)px * a.C + p.q * 4);
            const float d[4] = {v.x - K[0], v.y - K[1], v.z - K[2], v.w - K[3]};
#pragma unroll
            for (int e = 0; e < 4; ++e) s1[e] += d[e], s2[e] += d[e] * d[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[(threadIdx.x * 4 + e) * 2] = (double)s1[e];
            acc[(threadIdx.x * 4 + e) * 2 + 1] = (double)s2[e];
        }
        __syncthreads();
        if (threadIdx.x < a.G) {   // one thread per group: its cpg channels of every row
            double t1 = 0.0, t2 = 0.0;
            for (int r = 0; r < p.rows; ++r)
                for (int j = 0; j < cpg; ++j) {
                    const int c = threadIdx.x * cpg + j;
                    const int t = (r * qn + (c >> 2)) * 4 + (c & 3);
                    t1 += acc[t * 2];
                    t2 += acc[t * 2 + 1];
                }
            double *dst = a.sums + ((size_t)(L.img0 + p.b) * a.G + threadIdx.x) * 2;
            unsafeAtomicAdd(dst, t1);
            unsafeAtomicAdd(dst + 1, t2);
        }
        return;
    }
    const float K = xb[g * cpg];   // shift: first element of the group in this image
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 4
    for (int px = p.p0 + p.row; px < p.p1; px += p.rows) {
        const float4 v = *reinterpret_cast<const float4 *>(xb + (size_t)px * a.C + p.q * 4);
        const float d0 = v.x - K, d1 = v.y - K, d2 = v.z - K, d3 = v.w - K;
        s1 += (d0 + d1) + (d2 + d3);
        s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    // block reduction per group in fp64 through LDS
    acc[threadIdx.x * 2] = (double)s1;
    acc[threadIdx.x * 2 + 1] = (double)s2;
    __syncthreads();
    // threads of one group: quads q with (q*4)/cpg == g, all rows.  One thread per group sums them.
    const int qpg = cpg >> 2;
    if (threadIdx.x < a.G) {
        double t1 = 0.0, t2 = 0.0;
        for (int r = 0; r < p.rows; ++r)
            for (int j = 0; j < qpg; ++j) {
                const int t = r * qn + threadIdx.x * qpg + j;
                t1 += acc[t * 2];
                t2 += acc[t * 2 + 1];
            }
        double *dst = a.sums + ((size_t)(L.img0 + p.b) * a.G + threadIdx.x) * 2;
        unsafeAtomicAdd(dst, t1);
        unsafeAtomicAdd(dst + 1, t2);
    }
}

// mean / rstd of (image, group) from the shifted sums
__device__ __forceinline__ void gn_moments(const GnArgs &a, const GnLvl &L, int b, int g, float K, float &mean,
                                           float &rstd)
{
    const double n = (double)L.HW * (a.C / a.G);
    const double *s = a.sums + ((size_t)(L.img0 + b) * a.G + g) * 2;
    const double m1 = s[0] / n;
    double var = s[1] / n - m1 * m1;
    if (var < 0.0) var = 0.0;
    mean = (float)((double)K + m1);
    rstd = (float)(1.0 / sqrt(var + (double)a.eps));
}

// mean / rstd of the four channels of quad q: one group (EG = false) or one per element
template <bool EG>
__device__ __forceinline__ void gn_quad_moments_fwd(const GnArgs &a, const GnLvl &L, const float *xb, int b, int q, float (&mean)[4],
                                                    float (&rstd)[4])
{
    const int cpg = a.C / a.G;
    if constexpr (EG) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int g = (q * 4 + e) / cpg;
            gn_moments(a, L, b, g, xb[g * cpg], mean[e], rstd[e]);
        }
    } else {
        const int g = (q * 4) / cpg;
        gn_moments(a, L, b, g, xb[g * cpg], mean[0], rstd[0]);
        mean[1] = mean[2] = mean[3] = mean[0], rstd[1] = rstd[2] = rstd[3] = rstd[0];
    }
}
template <bool EG>
__device__ __forceinline__ void gn_quad_moments_saved(const GnArgs &a, const GnLvl &L, int b, int q, float (&mean)[4], float (&rstd)[4])
{
    const int cpg = a.C / a.G;
    const float *mr = a.mean_rstd + (size_t)(L.img0 + b) * a.G * 2;
    if constexpr (EG) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int g = (q * 4 + e) / cpg;
            mean[e] = mr[g * 2], rstd[e] = mr[g * 2 + 1];
        }
    } else {
        const int g = (q * 4) / cpg;
        mean[0] = mr[g * 2], rstd[0] = mr[g * 2 + 1];
        mean[1] = mean[2] = mean[3] = mean[0], rstd[1] = rstd[2] = rstd[3] = rstd[0];
    }
}

template <bool EG>
__global__ __launch_bounds__(256) void gn_apply_kernel(const GnArgs a)
{
    int li;
    const GnLvl &L = gn_level(a, blockIdx.x, li);
    const GnPos p = gn_pos(a, L);
    const int cpg = a.C / a.G;
    const float *xb = L.x + (size_t)p.b * L.HW * a.C;
    float *yb = L.y + (size_t)p.b * L.ybs;
    float mean[4], rstd[4];
    gn_quad_moments_fwd<EG>(a, L, xb, p.b, p.q, mean, rstd);
    if (p.p0 == 0 && p.row == 0) {   // saved for backward
#pragma unroll
        for (int e = 0; e < (EG ? 4 : 1); ++e)
            if ((p.q * 4 + e) % cpg == 0) {
                float *mr = a.mean_rstd + ((size_t)(L.img0 + p.b) * a.G + (p.q * 4 + e) / cpg) * 2;
                mr[0] = mean[e];
                mr[1] = rstd[e];
            }
    }
    const float4 ga = *reinterpret_cast<const float4 *>(a.gamma + p.q * 4);
    const float4 be = *reinterpret_cast<const float4 *>(a.beta + p.q * 4);
    const float a0 = rstd[0] * ga.x, a1 = rstd[1] * ga.y, a2 = rstd[2] * ga.z, a3 = rstd[3] * ga.w;
    const float b0 = be.x - mean[0] * a0, b1 = be.y - mean[1] * a1, b2 = be.z - mean[2] * a2, b3 = be.w - mean[3] * a3;
#pragma unroll 4
    for (int px = p.p0 + p.row; px < p.p1; px += p.rows) {
        const float4 v = *reinterpret_cast<const float4 *>(xb + (size_t)px * a.C + p.q * 4);
        float4 o = make_float4(v.x * a0 + b0, v.y * a1 + b1, v.z * a2 + b2, v.w * a3 + b3);
        if (a.relu) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
        *reinterpret_cast<float4 *>(yb + (size_t)px * a.C + p.q * 4) = o;
    }
}

// per (image, channel): A = sum dy' * xhat, Bc = sum dy'   (dy' = dy gated by the ReLU of the forward)
template <bool EG>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const GnArgs a)
{
    int li;
    const GnLvl &L = gn_level(a, blockIdx.x, li);
    const GnPos p = gn_pos(a, L);
    const float *xb = L.x + (size_t)p.b * L.HW * a.C;
    const float *db = L.dy + (size_t)p.b * L.dybs;
    float mean[4], rstd[4];
    gn_quad_moments_saved<EG>(a, L, p.b, p.q, mean, rstd);
    const float4 ga = *reinterpret_cast<const float4 *>(a.gamma + p.q * 4);
    const float4 be = *reinterpret_cast<const float4 *>(a.beta + p.q * 4);
    const float a0 = rstd[0] * ga.x, a1 = rstd[1] * ga.y, a2 = rstd[2] * ga.z, a3 = rstd[3] * ga.w;
    const float b0 = be.x - mean[0] * a0, b1 = be.y - mean[1] * a1, b2 = be.z - mean[2] * a2, b3 = be.w - mean[3] * a3;
    float A[4] = {0.f, 0.f, 0.f, 0.f}, Bc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int px = p.p0 + p.row; px < p.p1; px += p.rows) {
        const float4 v = *reinterpret_cast<const float4 *>(xb + (size_t)px * a.C + p.q * 4);
        float4 d = *reinterpret_cast<const float4 *>(db + (size_t)px * a.C + p.q * 4);
        const float h0 = (v.x - mean[0]) * rstd[0], h1 = (v.y - mean[1]) * rstd[1], h2 = (v.z - mean[2]) * rstd[2],
                    h3 = (v.w - mean[3]) * rstd[3];
        if (a.relu) {   // the forward's own expression, so the gate is bitwise the one that was applied
            d.x = (v.x * a0 + b0 > 0.f) ? d.x : 0.f;
            d.y = (v.y * a1 + b1 > 0.f) ? d.y : 0.f;
            d.z = (v.z * a2 + b2 > 0.f) ? d.z : 0.f;
            d.w = (v.w * a3 + b3 > 0.f) ? d.w : 0.f;
        }
        A[0] += d.x * h0, A[1] += d.y * h1, A[2] += d.z * h2, A[3] += d.w * h3;
        Bc[0] += d.x, Bc[1] += d.y, Bc[2] += d.z, Bc[3] += d.w;
    }
    __shared__ float red[256 * 8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[threadIdx.x * 8 + j] = A[j];
        red[threadIdx.x * 8 + 4 + j] = Bc[j];
    }
    __syncthreads();
    const int qn = a.C >> 2;
    if (threadIdx.x < qn) {   // row 0 threads sum the rows of their channel quad
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < p.rows; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += red[(r * qn + threadIdx.x) * 8 + j];
        // per-block partials, summed per image by gn_bwd_imgsum_kernel in a fixed order (round 2: fp32 atomics of up to
        // 263 blocks per address: contended, and different bits on every run)
        float *dst = a.part + ((size_t)blockIdx.x * a.C + threadIdx.x * 4) * 2;
        *reinterpret_cast<float4 *>(dst) = make_float4(s[0], s[4], s[1], s[5]);
        *reinterpret_cast<float4 *>(dst + 4) = make_float4(s[2], s[6], s[3], s[7]);
    }
}

// ab[image][c][2] = sum over the image's blocks of part[block][c][2].  grid = (image slots, 2 C / 32); the eight groups of
// 32 threads take every eighth block each and meet in LDS in a fixed order.
__global__ __launch_bounds__(256) void gn_bwd_imgsum_kernel(const GnArgs a, int images, int accumulate)
{
    __shared__ float red[8][32];
    const int im = blockIdx.x;
    int li = 0;
    while (li + 1 < a.nlv && im >= a.lv[li + 1].img0) ++li;
    const GnLvl &L = a.lv[li];
    const int tpi = (L.HW + GN_PIX - 1) / GN_PIX;
    const int t0 = L.tile0 + (im - L.img0) * tpi;
    const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int c2 = blockIdx.y * 32 + col;   // index into the 2 C interleaved (A, B) columns
    float s0 = 0.f, s1 = 0.f;
    if (c2 < 2 * a.C) {
        int b = grp;
        for (; b + 8 < tpi; b += 16) {
            s0 += a.part[(size_t)(t0 + b) * 2 * a.C + c2];
            s1 += a.part[(size_t)(t0 + b + 8) * 2 * a.C + c2];
        }
        if (b < tpi) s0 += a.part[(size_t)(t0 + b) * 2 * a.C + c2];
    }
    red[grp][col] = s0 + s1;
    __syncthreads();
    if (grp == 0 && c2 < 2 * a.C)
        store_agent(a.ab + (size_t)im * 2 * a.C + c2, ((red[0][col] + red[1][col]) + (red[2][col] + red[3][col])) +
                                                          ((red[4][col] + red[5][col]) + (red[6][col] + red[7][col])));
    // d gamma[c] = sum_images A[img][c], d beta[c] = sum_images Bc[img][c]: by the block that finishes last (round 5; a launch
    // of its own -- gn_param_grad_kernel -- until round 4: 16 launches of 5 us per step for 2 KB of output).  Hand-over
    // through agent-scope accesses, no fence (common.h).
    if (!a.dgamma && !a.dbeta) return;
    __shared__ unsigned last;
    wait_stores();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(a.ticket + 1, 1u) == gridDim.x * gridDim.y - 1;
    __syncthreads();
    if (!last) return;
    for (int c = threadIdx.x; c < a.C; c += 256) {
        float sa = 0.f, sb = 0.f;
        for (int i = 0; i < images; ++i) {
            sa += load_agent(a.ab + ((size_t)i * a.C + c) * 2);
            sb += load_agent(a.ab + ((size_t)i * a.C + c) * 2 + 1);
        }
        if (a.dgamma) a.dgamma[c] = accumulate ? a.dgamma[c] + sa : sa;
        if (a.dbeta) a.dbeta[c] = accumulate ? a.dbeta[c] + sb : sb;
    }
    if (threadIdx.x == 0) a.ticket[1] = 0;
}

template <bool EG>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const GnArgs a)
{
    int li;
    const GnLvl &L = gn_level(a, blockIdx.x, li);
    const GnPos p = gn_pos(a, L);
    const int cpg = a.C / a.G;
    const float *xb = L.x + (size_t)p.b * L.HW * a.C;
    const float *db = L.dy + (size_t)p.b * L.dybs;
    float *ob = L.dx + (size_t)p.b * L.HW * a.C;
    float mean[4], rstd[4];
    gn_quad_moments_saved<EG>(a, L, p.b, p.q, mean, rstd);
    // group sums S1 = sum_c gamma_c * Bc[c], S2 = sum_c gamma_c * A[c]
    const float inv_n = 1.f / ((float)L.HW * (float)cpg);
    float m1[4], m2[4];
#pragma unroll
    for (int e = 0; e < (EG ? 4 : 1); ++e) {
        const int g = (p.q * 4 + e) / cpg;
        float S1 = 0.f, S2 = 0.f;
        const float *ab = a.ab + ((size_t)(L.img0 + p.b) * a.C + g * cpg) * 2;
        for (int c = 0; c < cpg; ++c) {
            const float gm = a.gamma[g * cpg + c];
            S2 += gm * ab[c * 2];
            S1 += gm * ab[c * 2 + 1];
        }
        m1[e] = S1 * inv_n, m2[e] = S2 * inv_n;
    }
    if constexpr (!EG) m1[1] = m1[2] = m1[3] = m1[0], m2[1] = m2[2] = m2[3] = m2[0];
    const float4 ga = *reinterpret_cast<const float4 *>(a.gamma + p.q * 4);
    const float4 be = *reinterpret_cast<const float4 *>(a.beta + p.q * 4);
    const float a0 = rstd[0] * ga.x, a1 = rstd[1] * ga.y, a2 = rstd[2] * ga.z, a3 = rstd[3] * ga.w;
    const float b0 = be.x - mean[0] * a0, b1 = be.y - mean[1] * a1, b2 = be.z - mean[2] * a2, b3 = be.w - mean[3] * a3;
#pragma unroll 4
    for (int px = p.p0 + p.row; px < p.p1; px += p.rows) {
        const float4 v = *reinterpret_cast<const float4 *>(xb + (size_t)px * a.C + p.q * 4);
        float4 d = *reinterpret_cast<const float4 *>(db + (size_t)px * a.C + p.q * 4);
        const float h0 = (v.x - mean[0]) * rstd[0], h1 = (v.y - mean[1]) * rstd[1], h2 = (v.z - mean[2]) * rstd[2],
                    h3 = (v.w - mean[3]) * rstd[3];
        if (a.relu) {   // the forward's own expression, so the gate is bitwise the one that was applied
            d.x = (v.x * a0 + b0 > 0.f) ? d.x : 0.f;
            d.y = (v.y * a1 + b1 > 0.f) ? d.y : 0.f;
            d.z = (v.z * a2 + b2 > 0.f) ? d.z : 0.f;
            d.w = (v.w * a3 + b3 > 0.f) ? d.w : 0.f;
        }
        float4 o;
        o.x = rstd[0] * (ga.x * d.x - m1[0] - h0 * m2[0]);
        o.y = rstd[1] * (ga.y * d.y - m1[1] - h1 * m2[1]);
        o.z = rstd[2] * (ga.z * d.z - m1[2] - h2 * m2[2]);
        o.w = rstd[3] * (ga.w * d.w - m1[3] - h3 * m2[3]);
        *reinterpret_cast<float4 *>(ob + (size_t)px * a.C + p.q * 4) = o;
    }
}

// The statistics sums without a fill launch: two library-owned buffers per stream.  Call k accumulates into one of them and its
// statistics kernel clears what call k - 1 left in the other (stream order: the apply kernel of call k - 1, the only reader, has
// finished).  16 memsets per training step go; calls with more than GN_SUMS_CAP sums keep the caller's workspace + memset.
constexpr int GN_SUMS_CAP = 8192;   // doubles per buffer (images * G * 2)
struct GnSums {
    hipStream_t st;
    double *buf;   // [2][GN_SUMS_CAP], zeroed once
    int cur, dirty[2];
};
static GnSums g_gns[16];
static int g_ngns = 0;
static int gn_sums(hipStream_t st, int n, GnArgs &a)
{
    GnSums *s = nullptr;
    for (int i = 0; i < g_ngns; ++i)
        if (g_gns[i].st == st) s = &g_gns[i];
    if (!s) {
        if (g_ngns == 16) return fail(LSN_ERR_RUNTIME, "group norm: more than 16 streams use the library");
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
            return fail(LSN_ERR_RUNTIME, "group norm: first call inside a stream capture: run the step eagerly once before capturing");
        double *np = nullptr;
        LSN_HIP(hipMalloc(reinterpret_cast<void **>(&np), 2 * GN_SUMS_CAP * sizeof(double)));
        lib_stat(STAT_MALLOCS, 1), lib_stat(STAT_HELD_BYTES, (long long)(2 * GN_SUMS_CAP * sizeof(double)));
        LSN_HIP(hipMemsetAsync(np, 0, 2 * GN_SUMS_CAP * sizeof(double), st));
        s = &g_gns[g_ngns++];
        *s = GnSums{st, np, 0, {0, 0}};
    }
    const int cur = s->cur, oth = cur ^ 1;
    a.sums = s->buf + (size_t)cur * GN_SUMS_CAP;
    a.clear = s->buf + (size_t)oth * GN_SUMS_CAP;
    a.nclear = s->dirty[oth];
    s->dirty[oth] = 0, s->dirty[cur] = n, s->cur = oth;
    return 0;
}

static int gn_fill(GnArgs &a, int n, const lsn_gn_level *lv, int C, int G, int *tiles, int *images)
{
    LSN_CHECK(n >= 1 && n <= GN_MAXLV, "n_levels must be in [1,%d], got %d", GN_MAXLV, n);
    LSN_CHECK(C > 0 && G > 0 && C % G == 0, "num_channels %d must be divisible by num_groups %d", C, G);
    const int qn = C / 4;
    if (C % 4 != 0 || ((C / G) % 4 != 0 && 4 % (C / G) != 0) || qn > 256 || 256 % qn != 0 || G > 256)
        return fail(LSN_ERR_UNSUPPORTED, "group norm kernel needs C in {4..1024} with 256 %% (C/4) == 0 and "
                                         "C/G a multiple or a divisor of 4, got C=%d G=%d", C, G);
    int t = 0, im = 0;
    for (int i = 0; i < n; ++i) {
        LSN_CHECK(lv[i].B > 0 && lv[i].HW > 0 && lv[i].x != nullptr, "level %d: empty tensor", i);
        a.lv[i].x = lv[i].x;
        a.lv[i].y = lv[i].y;
        a.lv[i].dy = lv[i].dy;
        a.lv[i].dx = lv[i].dx;
        a.lv[i].B = lv[i].B;
        a.lv[i].HW = lv[i].HW;
        a.lv[i].ybs = lv[i].y_batch_stride ? lv[i].y_batch_stride : (long long)lv[i].HW * C;
        a.lv[i].dybs = lv[i].dy_batch_stride ? lv[i].dy_batch_stride : (long long)lv[i].HW * C;
        LSN_CHECK(a.lv[i].ybs >= (long long)lv[i].HW * C && a.lv[i].dybs >= (long long)lv[i].HW * C && a.lv[i].ybs % 4 == 0 &&
                      a.lv[i].dybs % 4 == 0,
                  "level %d: batch strides must cover an image and keep 16-byte alignment", i);
        a.lv[i].tile0 = t;
        a.lv[i].img0 = im;
        t += lv[i].B * ((lv[i].HW + GN_PIX - 1) / GN_PIX);
        im += lv[i].B;
    }
    a.nlv = n;
    a.C = C;
    a.G = G;
    *tiles = t;
    *images = im;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// BatchNorm with frozen statistics (eval mode: the LSNet backbones train with norm_eval=True, resnet.py:636-645)
// + optional residual add + optional ReLU, channels-last, one pass each way:
//   forward : y = act( x * a_c + b_c (+ res) ),  a_c = gamma_c / sqrt(var_c + eps),  b_c = beta_c - mean_c * a_c
//   backward: dz = dy * [y > 0];  dx = dz * a_c;  dres = dz;  dgamma_c = sum dz * (x - mean_c) * rstd_c;
//             dbeta_c = sum dz                      (ATen needs three kernels and five tensor reads for this)
// Pixel rows are C floats; a thread owns a channel quad for a strided set of pixels, block sums go through LDS and
// end in one fp32 atomic per channel and block.
// ---------------------------------------------------------------------------------------------------------
struct BnArgs {
    const float *x, *res, *dy, *y_in;
    float *y, *dx, *dres;
    const float *mean, *var, *gamma, *beta;
    float *dgamma, *dbeta;
    float *part;             // [blocks][2 C] per-block partial sums of (dgamma, dbeta)
    float eps;
    int N, C, relu;          // N = B * H * W pixels
};

constexpr int BN_PIX = 64;    // pixels per block (many small blocks: these kernels live on memory-level parallelism)

__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const BnArgs a)
{
    // blockIdx.y: 1024-channel slice (C > 1024: the 2048-channel maps of a ResNet's last stage)
    const int qb = blockIdx.y * 256, qn = min(256, (a.C >> 2) - qb);
    const int q = qb + threadIdx.x % qn, row = threadIdx.x / qn, rows = 256 / qn;
    const float4 mu = *reinterpret_cast<const float4 *>(a.mean + q * 4);
    const float4 va = *reinterpret_cast<const float4 *>(a.var + q * 4);
    const float4 ga = *reinterpret_cast<const float4 *>(a.gamma + q * 4);
    const float4 be = *reinterpret_cast<const float4 *>(a.beta + q * 4);
    const float a0 = ga.x * rsqrtf(va.x + a.eps), a1 = ga.y * rsqrtf(va.y + a.eps), a2 = ga.z * rsqrtf(va.z + a.eps),
                a3 = ga.w * rsqrtf(va.w + a.eps);
    const float b0 = be.x - mu.x * a0, b1 = be.y - mu.y * a1, b2 = be.z - mu.z * a2, b3 = be.w - mu.w * a3;
    const int p0 = blockIdx.x * BN_PIX, p1 = min(p0 + BN_PIX, a.N);
#pragma unroll 4
    for (int px = p0 + row; px < p1; px += rows) {
        const size_t o = (size_t)px * a.C + q * 4;
        const float4 v = *reinterpret_cast<const float4 *>(a.x + o);
        float4 r = make_float4(v.x * a0 + b0, v.y * a1 + b1, v.z * a2 + b2, v.w * a3 + b3);
        if (a.res) {
            const float4 e = *reinterpret_cast<const float4 *>(a.res + o);
            r.x += e.x, r.y += e.y, r.z += e.z, r.w += e.w;
        }
        if (a.relu) r = make_float4(fmaxf(r.x, 0.f), fmaxf(r.y, 0.f), fmaxf(r.z, 0.f), fmaxf(r.w, 0.f));
        *reinterpret_cast<float4 *>(a.y + o) = r;
    }
}

__global__ __launch_bounds__(256) void bn_act_bwd_kernel(const BnArgs a)
{
    // blockIdx.y: 1024-channel slice (C > 1024: the 2048-channel maps of a ResNet's last stage)
    const int qb = blockIdx.y * 256, qn = min(256, (a.C >> 2) - qb);
    const int q = qb + threadIdx.x % qn, row = threadIdx.x / qn, rows = 256 / qn;
    const float4 mu = *reinterpret_cast<const float4 *>(a.mean + q * 4);
    const float4 va = *reinterpret_cast<const float4 *>(a.var + q * 4);
    const float4 ga = *reinterpret_cast<const float4 *>(a.gamma + q * 4);
    const float r0 = rsqrtf(va.x + a.eps), r1 = rsqrtf(va.y + a.eps), r2 = rsqrtf(va.z + a.eps),
                r3 = rsqrtf(va.w + a.eps);
    const float a0 = ga.x * r0, a1 = ga.y * r1, a2 = ga.z * r2, a3 = ga.w * r3;
    float sg[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
    const int p0 = blockIdx.x * BN_PIX, p1 = min(p0 + BN_PIX, a.N);
#pragma unroll 4
    for (int px = p0 + row; px < p1; px += rows) {
        const size_t o = (size_t)px * a.C + q * 4;
        float4 d = *reinterpret_cast<const float4 *>(a.dy + o);
        float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.relu) y = *reinterpret_cast<const float4 *>(a.y_in + o);
        if (a.relu) {
            d.x = y.x > 0.f ? d.x : 0.f, d.y = y.y > 0.f ? d.y : 0.f, d.z = y.z > 0.f ? d.z : 0.f,
            d.w = y.w > 0.f ? d.w : 0.f;
        }
        if (a.dres) *reinterpret_cast<float4 *>(a.dres + o) = d;
        if (a.dx) *reinterpret_cast<float4 *>(a.dx + o) = make_float4(d.x * a0, d.y * a1, d.z * a2, d.w * a3);
        if (a.dgamma) {
            float4 v = *reinterpret_cast<const float4 *>(a.x + o);
            v.x -= mu.x, v.y -= mu.y, v.z -= mu.z, v.w -= mu.w;
            sg[0] += d.x * v.x, sg[1] += d.y * v.y, sg[2] += d.z * v.z, sg[3] += d.w * v.w;
            sb[0] += d.x, sb[1] += d.y, sb[2] += d.z, sb[3] += d.w;
        }
    }
    if (!a.dgamma) return;
    __shared__ float red[256 * 8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[threadIdx.x * 8 + j] = sg[j];
        red[threadIdx.x * 8 + 4 + j] = sb[j];
    }
    __syncthreads();
    if (threadIdx.x < qn) {
        float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < rows; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] += red[(r * qn + threadIdx.x) * 8 + j];
        // per-block partials, summed by bn_param_reduce_kernel: thousands of blocks adding atomically to the same
        // 2 C addresses cost more than the streaming pass itself (C = 64: 230 us against 50 us of traffic)
        float *dst = a.part + (size_t)blockIdx.x * 2 * a.C + (qb + threadIdx.x) * 4;
        *reinterpret_cast<float4 *>(dst) = make_float4(t[0] * r0, t[1] * r1, t[2] * r2, t[3] * r3);
        *reinterpret_cast<float4 *>(dst + a.C) = make_float4(t[4], t[5], t[6], t[7]);
    }
}

// dgamma[c] (+)= sum_blocks part[b][c], dbeta[c] (+)= sum_blocks part[b][C + c].  A workgroup owns 32 of the 2 C partial
// columns; its eight groups of 32 threads take every eighth block each and meet in LDS in a fixed order: no atomics, the
// same bits on every run (round 2 split the block range over grid.y and met in fp32 atomics).
__global__ __launch_bounds__(256) void bn_param_reduce_kernel(const BnArgs a, int blocks, int accumulate)
{
    __shared__ float red[8][32];
    const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + col;   // index into the 2 C partial columns
    float s0 = 0.f, s1 = 0.f;
    if (c < 2 * a.C) {
        int b = grp;
        for (; b + 8 < blocks; b += 16) {
            s0 += a.part[(size_t)b * 2 * a.C + c];
            s1 += a.part[(size_t)(b + 8) * 2 * a.C + c];
        }
        if (b < blocks) s0 += a.part[(size_t)b * 2 * a.C + c];
    }
    red[grp][col] = s0 + s1;
    __syncthreads();
    if (grp == 0 && c < 2 * a.C) {
        const float s = ((red[0][col] + red[1][col]) + (red[2][col] + red[3][col])) +
                        ((red[4][col] + red[5][col]) + (red[6][col] + red[7][col]));
        float *dst = (c < a.C ? a.dgamma : a.dbeta - a.C) + c;
        *dst = accumulate ? *dst + s : s;
    }
}

// grad = grad_y where y > 0, else 0 (the gradient through a ReLU whose output y is what was stored)
__global__ __launch_bounds__(256) void relu_gate_kernel(const float4 *__restrict__ gy, const float4 *__restrict__ y,
                                                        float4 *__restrict__ g, int64_t n4)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 d = gy[i], v = y[i];
        g[i] = make_float4(v.x > 0.f ? d.x : 0.f, v.y > 0.f ? d.y : 0.f, v.z > 0.f ? d.z : 0.f, v.w > 0.f ? d.w : 0.f);
    }
}

// the same over several tensors (the FPN levels of one multi-level convolution) in one launch; grad_y may keep its images apart
// (a slice of a concatenated gradient: lsn_gate_job.gy_batch_stride)
constexpr int GATE_MAX_JOBS = 8;
struct GateJob {
    const float4 *gy, *y;
    float4 *g;
    long long per4, gy_stride4, n4;   // float4 per image, between the images of gy, in all
    int blk0;                         // first block; a block gates 1024 float4
};
struct GateJobs {
    GateJob j[GATE_MAX_JOBS];
    int n;
};
__global__ __launch_bounds__(256) void relu_gate_multi_kernel(const GateJobs a)
{
    int k = 0;
    while (k + 1 < a.n && (int)blockIdx.x >= a.j[k + 1].blk0) ++k;
    const GateJob &J = a.j[k];
    const long long base = (long long)(blockIdx.x - J.blk0) * 1024;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long i = base + u * 256 + threadIdx.x;
        if (i >= J.n4) break;
        const long long b = i / J.per4, r = i - b * J.per4;
        const float4 d = J.gy[b * J.gy_stride4 + r], v = J.y[i];
        J.g[i] = make_float4(v.x > 0.f ? d.x : 0.f, v.y > 0.f ? d.y : 0.f, v.z > 0.f ? d.z : 0.f, v.w > 0.f ? d.w : 0.f);
    }
}

static int bn_check(int N, int C);

// shared tail of the backward entry point; `reads`: tensors of N x C floats the kernel reads
static int bn_backward_launch(BnArgs &a, void *workspace, int accumulate, int reads, hipStream_t st)
{
    const int N = a.N, C = a.C;
    const int blocks = (N + BN_PIX - 1) / BN_PIX;
    if (a.dgamma) {
        LSN_CHECK(workspace != nullptr, "batch norm backward: grad_gamma needs the workspace");
        a.part = reinterpret_cast<float *>(workspace);
    }
    ProfSpan prof(PROF_NORM, 6.0 * N * C, 4.0 * N * C * (reads + (a.dx ? 1 : 0) + (a.dres ? 1 : 0)), st);
    hipLaunchKernelGGL(bn_act_bwd_kernel, dim3(blocks, (C / 4 + 255) / 256), dim3(256), 0, st, a);
    if (a.dgamma)
        hipLaunchKernelGGL(bn_param_reduce_kernel, dim3((2 * C + 31) / 32), dim3(256), 0, st, a, blocks, accumulate ? 1 : 0);
    LSN_HIP(hipGetLastError());
    return 0;
}

static int bn_check(int N, int C)
{
    LSN_CHECK(N > 0 && C > 0, "batch norm: empty tensor");
    const int qn = C / 4;
    if (C % 4 != 0 || (qn <= 256 ? 256 % qn != 0 : qn % 256 != 0))
        return fail(LSN_ERR_UNSUPPORTED, "batch norm kernel needs C in {4..1024} with 256 %% (C/4) == 0, or a multiple of 1024, got %d", C);
    return 0;
}

}  // namespace lsn

extern "C" {

int64_t lsn_group_norm_workspace_bytes(int n_levels, const lsn_gn_level *levels, int C, int G)
{
    int64_t images = 0;
    for (int i = 0; i < n_levels; ++i) images += levels[i].B;
    int64_t tiles = 0;
    for (int i = 0; i < n_levels; ++i) tiles += (int64_t)levels[i].B * ((levels[i].HW + lsn::GN_PIX - 1) / lsn::GN_PIX);
    // forward: sums (double [images][G][2]); backward: ab (float [images][C][2]) + per-block partials (float
    // [blocks][C][2]); sized for the larger
    const int64_t f = images * G * 2 * (int64_t)sizeof(double), b = (images + tiles) * C * 2 * (int64_t)sizeof(float);
    return f > b ? f : b;
}

int lsn_group_norm_forward(int n_levels, const lsn_gn_level *levels, int C, int G, const float *gamma,
                           const float *beta, float eps, int relu, float *mean_rstd, void *workspace,
                           lsn_stream_t stream)
{
    using namespace lsn;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    GnArgs a = {};
    int tiles = 0, images = 0;
    if (int rc = gn_fill(a, n_levels, levels, C, G, &tiles, &images)) return rc;
    LSN_CHECK(gamma && beta && mean_rstd && workspace, "group norm: NULL argument");
    for (int i = 0; i < n_levels; ++i) LSN_CHECK(levels[i].y != nullptr, "level %d: output is NULL", i);
    a.gamma = gamma;
    a.beta = beta;
    a.eps = eps;
    a.relu = relu;
    a.mean_rstd = mean_rstd;
    // (the two alternating buffers are HOST state baked into the launch -- which buffer this call sums into, how much
    // of the other it clears.  A captured graph replays the capture-time choice, which is right only while replays and eager
    // calls keep the parity the capture saw; a capturing stream therefore takes the self-contained form: the caller's
    // workspace and a memset node.)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    if (images * G * 2 <= GN_SUMS_CAP && !capturing) {
        if (int rc = gn_sums(st, images * G * 2, a)) return rc;
    } else {
        a.sums = reinterpret_cast<double *>(workspace);
        LSN_HIP(hipMemsetAsync(a.sums, 0, sizeof(double) * (size_t)images * G * 2, st));
    }
    double el = 0;
    for (int i = 0; i < n_levels; ++i) el += (double)levels[i].B * levels[i].HW * C;
    ProfSpan prof(PROF_NORM, 8.0 * el, 4.0 * 2 * el, st);   // algorithmic: x read once, y written once
    if ((C / G) % 4 != 0) {   // one or two channels per group
        hipLaunchKernelGGL(gn_stats_kernel<true>, dim3(tiles), dim3(256), 0, st, a);
        hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(tiles), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL(gn_stats_kernel<false>, dim3(tiles), dim3(256), 0, st, a);
        hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(tiles), dim3(256), 0, st, a);
    }
    LSN_HIP(hipGetLastError());
    return 0;
}

int lsn_group_norm_backward(int n_levels, const lsn_gn_level *levels, int C, int G, const float *gamma,
                            const float *beta, int relu, const float *mean_rstd, float *grad_gamma,
                            float *grad_beta, void *workspace, int accumulate, lsn_stream_t stream)
{
    using namespace lsn;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    GnArgs a = {};
    int tiles = 0, images = 0;
    if (int rc = gn_fill(a, n_levels, levels, C, G, &tiles, &images)) return rc;
    LSN_CHECK(gamma && beta && mean_rstd && workspace, "group norm: NULL argument");
    for (int i = 0; i < n_levels; ++i)
        LSN_CHECK(levels[i].dy != nullptr && levels[i].dx != nullptr, "level %d: dy/dx is NULL", i);
    a.gamma = gamma;
    a.beta = beta;
    a.relu = relu;
    a.mean_rstd = const_cast<float *>(mean_rstd);
    a.ab = reinterpret_cast<float *>(workspace);
    a.part = a.ab + (size_t)images * C * 2;
    a.dgamma = grad_gamma;
    a.dbeta = grad_beta;
    if (int rc = lib_tickets(&a.ticket, st)) return rc;
    double el = 0;
    for (int i = 0; i < n_levels; ++i) el += (double)levels[i].B * levels[i].HW * C;
    ProfSpan prof(PROF_NORM, 16.0 * el, 4.0 * 3 * el, st);   // x, dy read, dx written
    const bool eg = (C / G) % 4 != 0;   // one or two channels per group
    if (eg)
        hipLaunchKernelGGL(gn_bwd_reduce_kernel<true>, dim3(tiles), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(gn_bwd_reduce_kernel<false>, dim3(tiles), dim3(256), 0, st, a);
    hipLaunchKernelGGL(gn_bwd_imgsum_kernel, dim3(images, (2 * C + 31) / 32), dim3(256), 0, st, a, images, accumulate);
    if (eg)
        hipLaunchKernelGGL(gn_bwd_apply_kernel<true>, dim3(tiles), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(gn_bwd_apply_kernel<false>, dim3(tiles), dim3(256), 0, st, a);
    LSN_HIP(hipGetLastError());
    return 0;
}

int lsn_bn_eval_act_forward(const float *x, const float *residual, float *y, const float *running_mean,
                            const float *running_var, const float *gamma, const float *beta, float eps, int relu,
                            int N, int C, lsn_stream_t stream)
{
    using namespace lsn;
    if (int rc = bn_check(N, C)) return rc;
    LSN_CHECK(x && y && running_mean && running_var && gamma && beta, "batch norm: NULL argument");
    BnArgs a = {};
    a.x = x, a.res = residual, a.y = y, a.mean = running_mean, a.var = running_var, a.gamma = gamma, a.beta = beta;
    a.eps = eps, a.N = N, a.C = C, a.relu = relu;
    ProfSpan prof(PROF_NORM, 4.0 * N * C, 4.0 * N * C * (residual ? 3 : 2), reinterpret_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3((N + BN_PIX - 1) / BN_PIX, (C / 4 + 255) / 256), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), a);
    LSN_HIP(hipGetLastError());
    return 0;
}

int64_t lsn_bn_eval_act_workspace_bytes(int N, int C)
{
    return (int64_t)((N + lsn::BN_PIX - 1) / lsn::BN_PIX) * 2 * C * (int64_t)sizeof(float);
}

int lsn_bn_eval_act_backward(const float *grad_y, const float *y, const float *x, const float *running_mean,
                             const float *running_var, const float *gamma, float eps, int relu, float *grad_x,
                             float *grad_residual, float *grad_gamma, float *grad_beta, void *workspace, int N,
                             int C, int accumulate, lsn_stream_t stream)
{
    using namespace lsn;
    if (int rc = bn_check(N, C)) return rc;
    LSN_CHECK(grad_y && running_mean && running_var && gamma, "batch norm: NULL argument");
    LSN_CHECK(!relu || y, "batch norm backward: the ReLU gate needs y");
    LSN_CHECK((grad_gamma == nullptr) == (grad_beta == nullptr), "grad_gamma and grad_beta come together");
    LSN_CHECK(!grad_gamma || x, "batch norm backward: grad_gamma needs x");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    BnArgs a = {};
    a.dy = grad_y, a.y_in = y, a.x = x, a.mean = running_mean, a.var = running_var, a.gamma = gamma;
    a.dx = grad_x, a.dres = grad_residual, a.dgamma = grad_gamma, a.dbeta = grad_beta;
    a.eps = eps, a.N = N, a.C = C, a.relu = relu;
    return bn_backward_launch(a, workspace, accumulate, 1 + (relu ? 1 : 0) + (grad_gamma ? 1 : 0), st);
}

int lsn_relu_gate(const float *grad_y, const float *y, float *grad, int64_t n, lsn_stream_t stream)
{
    using namespace lsn;
    LSN_CHECK(grad_y && y && grad && n > 0 && n % 4 == 0, "relu gate: bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ProfSpan prof(PROF_NORM, 1.0 * n, 12.0 * n, st);
    const int64_t n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 16384 ? (n4 + 255) / 256 : 16384);
    hipLaunchKernelGGL(relu_gate_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const float4 *>(grad_y),
                       reinterpret_cast<const float4 *>(y), reinterpret_cast<float4 *>(grad), n4);
    LSN_HIP(hipGetLastError());
    return 0;
}

int lsn_relu_gate_multi(int n_jobs, const lsn_gate_job *jobs, lsn_stream_t stream)
{
    using namespace lsn;
    LSN_CHECK(n_jobs >= 1 && n_jobs <= GATE_MAX_JOBS && jobs, "relu gate: 1 .. %d tensors per launch", GATE_MAX_JOBS);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    GateJobs a = {};
    a.n = n_jobs;
    int nb = 0;
    double el = 0;
    for (int k = 0; k < n_jobs; ++k) {
        const lsn_gate_job &q = jobs[k];
        LSN_CHECK(q.grad_y && q.y && q.grad && q.B >= 1 && q.per_image > 0 && q.per_image % 4 == 0 && q.gy_batch_stride % 4 == 0 &&
                      q.gy_batch_stride >= q.per_image, "relu gate: tensor %d: bad arguments", k);
        LSN_CHECK(((reinterpret_cast<uintptr_t>(q.grad_y) | reinterpret_cast<uintptr_t>(q.y) | reinterpret_cast<uintptr_t>(q.grad)) & 15) == 0,
                  "relu gate: tensor %d is not 16-byte aligned", k);
        GateJob &J = a.j[k];
        J.gy = reinterpret_cast<const float4 *>(q.grad_y), J.y = reinterpret_cast<const float4 *>(q.y);
        J.g = reinterpret_cast<float4 *>(q.grad);
        J.per4 = q.per_image / 4, J.gy_stride4 = q.gy_batch_stride / 4, J.n4 = J.per4 * q.B, J.blk0 = nb;
        nb += (int)((J.n4 + 1023) / 1024);
        el += 4.0 * J.n4;
    }
    ProfSpan prof(PROF_NORM, el, 12.0 * el, st);
    hipLaunchKernelGGL(relu_gate_multi_kernel, dim3(nb), dim3(256), 0, st, a);
    LSN_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
