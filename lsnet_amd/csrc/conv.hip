// Dense 2-D convolution (groups = 1) on channels-last fp32 tensors as an implicit GEMM on the bf16 matrix pipe with
// split operands (common.h: NP = 6 products of exact 3-way bf16 splits = fp32-equivalent, the default; NP = 3 products
// of 2-way splits), fp32 accumulation.
//
// The reference runs torch.nn.Conv2d = cuDNN for every dense conv of the path (backbone resnet.py:624-631,
// 261-301; neck fpn.py:171-217; head lsnet_head.py:160-257).  On MI355X fp32 MFMA runs at the fp32 vector rate
// (157 TF, 1/16 of the bf16 rate), so an exact-fp32 GEMM tops out near 100 TF in practice (the vendor igemm kernels
// measure 38-117 TF on these shapes, tools/bench_convs.py).  This kernel keeps fp32 tensors in memory and splits both
// operands into bf16 planes while staging them into LDS (weights once per call: conv_prepare_kernel).
//
//   forward      : out[p][co] = sum_{tap, ci} x[p @ tap][ci] * w[co][tap][ci] (+ bias, ReLU); up to 8 maps per launch
//   backward-data: the same kernel on grad_output with the weights transposed and flipped; a stride-s convolution is
//                  s x s stride-1 convolutions over tap subsets (TapSub), each writing its residue class of input pixels
//   backward-weight: dcn_wgrad_xn_kernel<PLAIN = true> (dcn_kernels.h), reduction over pixels with px-contiguous LDS images
//
// Tiling: block = BM output pixels x BN output channels, BM + BN = 320, four waves each owning a 64x64 tile
// (1x4, 2x2 or 4x1 waves); chunk = one tap x 32 input channels; software pipeline over chunks exactly as
// dcn_fwd_xn_kernel: MFMAs of chunk t, LDS commit of chunk t+1, global-load issue of chunk t+2, one staging
// slice in every MFMA gap.  LDS rows are 32 bf16 + 16 B pad (80 B) so the 16-byte operand reads are conflict free.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lsnet_hip.h"
#include "common.h"

namespace lsn {

constexpr int CV_MAXLV = 8;

// One input map of a batched launch: the FPN levels that share a convolution's weights (LSHead) go into ONE launch,
// so that the small levels do not each pay a launch whose duration is set by the depth of the reduction.
struct ConvLvl {
    const float *x;
    float *out;
    int B, H, W, Ho, Wo;
    int P;       // B * Ho * Wo
    int tile0;   // first pixel tile of this level
};

struct ConvArgs {
    ConvLvl lv[CV_MAXLV];
    int nlv, ntiles;
    const float *w, *bias;
    int C, Co, kh, kw, stride, pad_h, pad_w, dil;
    int xpitch;   // floats between horizontally adjacent input pixels (= C, except for the row-merged stem form)
    int relu;
    int ksplit;   // > 1: the chunk range is divided over blockIdx.z and the partial sums are added atomically into a
                  // zero-filled output (few pixels, deep reduction: FPN P6 / P7); bias by split 0, no ReLU
    const unsigned short *wp;   // prepared weights: NPL bf16 planes [Co][K][C], or NULL
    // output placement: pixel (b, ho, wo) of the (Ho, Wo) grid is stored at (b, oy0 + ho * ostep, ox0 + wo * ostep) of an
    // (OH, OW) map.  ostep = 0: the dense case (OH = Ho, OW = Wo).  Used by the strided backward-data pass, which
    // computes each residue class of input pixels as its own stride-1 convolution over a subset of the taps.
    int ostep, oy0, ox0, OH, OW;
    long long *dbg;   // optional phase timestamps of block dbg_block, thread 0 (lsn_debug_phase_clocks)
    int dbg_block;
};

#define CV_STAMP(slot)                                                                                            \
    do {                                                                                                          \
        if (a.dbg != nullptr && blockIdx.x == (unsigned)(a.dbg_block & 0xfffff) && blockIdx.y == 0 &&             \
            blockIdx.z == 0 && threadIdx.x == 0 && dbg_n < 512)                                                    \
            a.dbg[dbg_n++] = ((long long)(slot) << 56) | (clock64() & 0x00ffffffffffffffll);                       \
    } while (0)

constexpr int CV_RS = 80;   // LDS row stride in bytes

__device__ __forceinline__ float2 cv_load2(__amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    auto v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
    float2 f;
    __builtin_memcpy(&f, &v, 8);
    return f;
}
__device__ __forceinline__ float4 cv_load4(__amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    float4 f;
    __builtin_memcpy(&f, &v, 16);
    return f;
}

// PREP: the weights arrive already split (conv_prepare_kernel): their staging is a 16-byte copy per slice.
// NP: bf16 products per fp32 product (common.h): 3 on two planes, or 6 on three planes (fp32-equivalent).
template <int BM, int BN, bool PREP, int NP>
__global__ __launch_bounds__(256, 1) void conv_fwd_xn_kernel(const ConvArgs a)
{
    using SC = SplitCfg<NP>;
    constexpr int NPL = SC::NPL;
    constexpr int WM = BM / 64, WN = BN / 64, RS = CV_RS, BK = 32;
    static_assert(WM * WN == 4, "four waves of 64x64");
    constexpr int NPA = BM / 16, NPB = PREP ? NPL * (BN / 64) : BN / 32, NS = NPA + NPB;   // staging slices per chunk
    constexpr int PLANE_A = BM * RS, PLANE_B = BN * RS, BUF = NPL * (PLANE_A + PLANE_B);
    constexpr int NGAP = 2 * NP * 4;
    constexpr int NSLOT = 4 * NPA + 2 * NPB;   // micro-slots of a chunk's staging: 4 per pixel slice, 2 per weight slice
    extern __shared__ __align__(16) unsigned char smem[];   // 2 x [A planes][B planes]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int K = a.kh * a.kw, Kdim = K * a.C;
    // XCD-ordered (pixel tile, column block) with the column blocks of a pixel tile adjacent (same input rows)
    const int work = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int ptile = work / (int)gridDim.y;
    int li = 0;
    while (li + 1 < a.nlv && ptile >= a.lv[li + 1].tile0) ++li;
    const ConvLvl &L = a.lv[li];
    const int tile_p = (ptile - L.tile0) * BM;
    const int co_blk = (work - ptile * (int)gridDim.y) * BN;
    const int nco = min(BN, a.Co - co_blk);
    const int ncc = (a.C + BK - 1) / BK;
    const int Tall = K * ncc;
    // split-K: this block reduces chunks [t_begin, t_begin + T)
    const int t_begin = (int)((long long)Tall * blockIdx.z / gridDim.z);
    const int T = (int)((long long)Tall * (blockIdx.z + 1) / gridDim.z) - t_begin;

    const int kk2 = tid & 15, prow = tid >> 4;   // gather: channel pair, pixel row (16 rows per pass)
    const int wq = tid & 7, wrow = tid >> 3;     // weights: float4 slot along ci, co row (32 rows per pass)

    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(L.x), 0, L.B * L.H * L.W * a.xpitch * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs =
        PREP ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(a.wp), 0, a.Co * Kdim * 2 * NPL, 0x00020000)
             : __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.w), 0, a.Co * Kdim * 4, 0x00020000);
    constexpr int OOB = 0x7ffffff0;   // beyond num_records: the buffer load returns 0
    // PREP staging: slice ps covers plane ps % NPL, rows (ps / NPL) * 64 + (tid >> 2), 16-byte slot (tid & 3)
    const int pq = tid & 3, prw = tid >> 2;

    // per gather pass: top-left input coordinate of the pixel's receptive field and its image base
    int iy0[NPA], ix0[NPA], ibase[NPA];
#pragma unroll
    for (int ps = 0; ps < NPA; ++ps) {
        const int p = tile_p + ps * 16 + prow;
        const bool ok = p < L.P;
        const int HWo = L.Ho * L.Wo;
        const int b = ok ? p / HWo : 0, rem = ok ? p - b * HWo : 0;
        const int ho = rem / L.Wo, wo = rem - ho * L.Wo;
        iy0[ps] = ok ? ho * a.stride - a.pad_h : -0x40000000;   // an invalid pixel is out of range for every tap
        ix0[ps] = wo * a.stride - a.pad_w;
        ibase[ps] = b * L.H * L.W;
    }
    // For the pipelined loop: byte offset of the pass's pixel at tap (0, 0), channel pair kk2, and one validity bit per
    // tap (kh * kw <= 64: checked by the host).  A chunk then costs one add and one bit test per slice instead of the
    // coordinate arithmetic above.
    int pbase[NPA];
    unsigned long long vmask[NPA];
#pragma unroll
    for (int ps = 0; ps < NPA; ++ps) {
        pbase[ps] = ((ibase[ps] + iy0[ps] * L.W + ix0[ps]) * a.xpitch + 2 * kk2) * 4;
        unsigned long long m = 0;
        for (int i = 0; i < a.kh; ++i)
            for (int j = 0; j < a.kw; ++j) {
                const int y = iy0[ps] + i * a.dil, x = ix0[ps] + j * a.dil;
                if ((unsigned)y < (unsigned)L.H && (unsigned)x < (unsigned)L.W) m |= 1ull << (i * a.kw + j);
            }
        vmask[ps] = m;
    }
    int wvoff[NPB];
#pragma unroll
    for (int ps = 0; ps < NPB; ++ps) {
        if (PREP) {
            const int col = (ps / NPL) * 64 + prw;
            wvoff[ps] = (col < nco) ? ((ps % NPL) * a.Co * Kdim + (co_blk + col) * Kdim) * 2 + pq * 16 : OOB;
        } else {
            const int col = ps * 32 + wrow;
            wvoff[ps] = (col < nco) ? ((co_blk + col) * Kdim + wq * 4) * 4 : OOB;
        }
    }

    float2 xv[NPA];
    float4 wv[NPB];
    // chunk t = (tap k, channel slab cc): k = t / ncc walked incrementally
    struct Ck {
        int i, j, cc;
    };
    auto next = [&](Ck &c) {
        if (++c.cc == ncc) {
            c.cc = 0;
            if (++c.j == a.kw) {
                c.j = 0;
                ++c.i;
            }
        }
    };
    auto issue_x = [&](const Ck &c, int ps) {
        const int y = iy0[ps] + c.i * a.dil, x = ix0[ps] + c.j * a.dil;
        const bool ok = (unsigned)y < (unsigned)L.H && (unsigned)x < (unsigned)L.W && c.cc * BK + 2 * kk2 < a.C;
        const int voff = ok ? ((ibase[ps] + y * L.W + x) * a.xpitch + c.cc * BK + 2 * kk2) * 4 : OOB;
        xv[ps] = cv_load2(xrs, voff, 0);
    };
    auto issue_w = [&](const Ck &c, int ps) {
        if (PREP) {   // columns past C hold the next tap's values; the A operand is zero there
            wv[ps] = cv_load4(wrs, wvoff[ps], ((c.i * a.kw + c.j) * a.C + c.cc * BK) * 2);
        } else {
            const bool ok = c.cc * BK + wq * 4 < a.C;
            wv[ps] = cv_load4(wrs, ok ? wvoff[ps] : OOB, ((c.i * a.kw + c.j) * a.C + c.cc * BK) * 4);
        }
    };
    auto commit_x = [&](int ps, unsigned char *buf) {
        unsigned pl[NPL];
        split_planes<NPL>(xv[ps].x, xv[ps].y, pl);
        unsigned char *p = buf + (ps * 16 + prow) * RS + kk2 * 4;
#pragma unroll
        for (int q = 0; q < NPL; ++q) *reinterpret_cast<unsigned *>(p + q * PLANE_A) = pl[q];
    };
    auto commit_w = [&](int ps, unsigned char *buf) {
        if (PREP) {
            unsigned char *p = buf + NPL * PLANE_A + (ps % NPL) * PLANE_B + ((ps / NPL) * 64 + prw) * RS + pq * 16;
            *reinterpret_cast<float4 *>(p) = wv[ps];
            return;
        }
        unsigned p0[NPL], p1[NPL];
        split_planes<NPL>(wv[ps].x, wv[ps].y, p0);
        split_planes<NPL>(wv[ps].z, wv[ps].w, p1);
        unsigned char *p = buf + NPL * PLANE_A + (ps * 32 + wrow) * RS + wq * 8;
#pragma unroll
        for (int q = 0; q < NPL; ++q) *reinterpret_cast<uint2 *>(p + q * PLANE_B) = make_uint2(p0[q], p1[q]);
    };

    // ---- the same staging cut into micro-slots of <= ~6 instructions for the MFMA gaps of the pipelined loop ----------
    // A wave owns its SIMD alone (one workgroup per CU), so whatever does not fit into the 32-cycle shadow of an MFMA
    // stalls the matrix pipe: tools/phase_clocks.py conv measured 1.64 k cycles for the 48 MFMAs of a chunk alone and
    // 2.8 k with one whole (commit + issue) slice per gap.  Pixel slice ps = 4 slots: split step A (hi plane + residual),
    // split step B (mid / lo planes), the LDS writes, address + load of chunk t+2; weight slice = 2 slots.
    unsigned sp_h = 0, sp_m = 0, sp_l = 0;
    float sp_r0 = 0.f, sp_r1 = 0.f;
    int tap2 = 0, toff2 = 0, clim2 = 0;   // chunk t+2: tap index, byte offset of (tap, channel slab), valid channels
    auto chunk_scalars = [&](const Ck &c) {
        tap2 = c.i * a.kw + c.j;
        toff2 = ((c.i * a.dil * L.W + c.j * a.dil) * a.xpitch + c.cc * BK) * 4;
        clim2 = a.C - c.cc * BK;
    };
    auto micro_x = [&](int ps, int part, unsigned char *buf) {
        if (part == 0) {
            const bf16x2 h = {(__bf16)xv[ps].x, (__bf16)xv[ps].y};
            sp_h = __builtin_bit_cast(unsigned, h);
            sp_r0 = xv[ps].x - __uint_as_float(sp_h << 16);
            sp_r1 = xv[ps].y - __uint_as_float(sp_h & 0xffff0000u);
        } else if (part == 1) {
            const bf16x2 m = {(__bf16)sp_r0, (__bf16)sp_r1};
            sp_m = __builtin_bit_cast(unsigned, m);
            if constexpr (NPL == 3) {
                const float s0 = sp_r0 - __uint_as_float(sp_m << 16), s1 = sp_r1 - __uint_as_float(sp_m & 0xffff0000u);
                const bf16x2 l = {(__bf16)s0, (__bf16)s1};
                sp_l = __builtin_bit_cast(unsigned, l);
            }
        } else if (part == 2) {
            unsigned char *p = buf + (ps * 16 + prow) * RS + kk2 * 4;
            *reinterpret_cast<unsigned *>(p) = sp_h;
            *reinterpret_cast<unsigned *>(p + PLANE_A) = sp_m;
            if constexpr (NPL == 3) *reinterpret_cast<unsigned *>(p + 2 * PLANE_A) = sp_l;
        } else {
            const bool ok = ((vmask[ps] >> tap2) & 1ull) != 0 && 2 * kk2 < clim2;
            xv[ps] = cv_load2(xrs, ok ? pbase[ps] + toff2 : OOB, 0);
        }
    };

    // leading product h*h in acc, the small products in accl (added once at the end): the fp32 rounding of the large
    // running sum is then paid once per 16 k-values, not once per product term
    f32x16 acc[2][2], accl[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = accl[i][j][r] = 0.f;

    // ---- prologue: first chunk -> buffer 0, loads of the second in flight ----
    Ck c1;
    {
        const int k0 = t_begin / ncc;
        c1.cc = t_begin - k0 * ncc;
        c1.i = k0 / a.kw;
        c1.j = k0 - c1.i * a.kw;
    }
    Ck c2 = c1;
    {
#pragma unroll
        for (int ps = 0; ps < NPA; ++ps) issue_x(c1, ps);
#pragma unroll
        for (int ps = 0; ps < NPB; ++ps) issue_w(c1, ps);
#pragma unroll
        for (int ps = 0; ps < NPA; ++ps) commit_x(ps, smem);
#pragma unroll
        for (int ps = 0; ps < NPB; ++ps) commit_w(ps, smem);
        if (T > 1) next(c1);
        c2 = c1;
        if (T > 2) next(c2);
#pragma unroll
        for (int ps = 0; ps < NPA; ++ps) issue_x(c1, ps);
#pragma unroll
        for (int ps = 0; ps < NPB; ++ps) issue_w(c1, ps);
    }
    __syncthreads();

    int dbg_n = 0;
    for (int t = 0; t < T; ++t) {
        CV_STAMP(2);
        const int cur = t & 1;
        const unsigned char *bc = smem + cur * BUF;
        unsigned char *bn = smem + (cur ^ 1) * BUF;
        // registers hold chunk t+1 (to commit); c2 = chunk t+2 (to issue); both saturate at the last chunk
        const unsigned char *ap = bc + (wm * 64 + (lane & 31)) * RS + (lane >> 5) * 16;
        const unsigned char *bp = bc + NPL * PLANE_A + (wn * 64 + (lane & 31)) * RS + (lane >> 5) * 16;
        chunk_scalars(c2);
        bf16x8 Af[2][2][NPL], Bf[2][2][NPL];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    Af[ks][i][q] = *reinterpret_cast<const bf16x8 *>(ap + q * PLANE_A + i * 32 * RS + ks * 32);
                    Bf[ks][i][q] = *reinterpret_cast<const bf16x8 *>(bp + q * PLANE_B + i * 32 * RS + ks * 32);
                }
        if (a.dbg != nullptr) {
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): diagnostic only (operand reads landed)
            CV_STAMP(5);
        }
        // NGAP MFMAs, NS staging slice pairs (commit, issue) spread over the gaps
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int prod = 0; prod < NP; ++prod)
#pragma unroll
                for (int ij = 0; ij < 4; ++ij) {
                    const int i = ij >> 1, j = ij & 1;
                    const int gap = (ks * NP + prod) * 4 + ij;
                    if (prod == 0)
                        acc[i][j] = mfma_bf16(Af[ks][i][0], Bf[ks][j][0], acc[i][j]);
                    else
                        accl[i][j] = mfma_bf16(Af[ks][i][SC::pa(prod)], Bf[ks][j][SC::pb(prod)], accl[i][j]);
                    __builtin_amdgcn_sched_barrier(0);
                    // micro-slots [gap * NSLOT / NGAP, (gap + 1) * NSLOT / NGAP) of the staging (see micro_x)
#ifndef CV_ABLATE_STAGING   // (diagnostic builds: the MFMA block alone, tools/phase_clocks.py conv)
#pragma unroll
                    for (int s = gap * NSLOT / NGAP; s < (gap + 1) * NSLOT / NGAP; ++s) {
                        if (s < 4 * NPA) {
                            micro_x(s >> 2, s & 3, bn);
                        } else {
                            const int sw = s - 4 * NPA;
                            if ((sw & 1) == 0) commit_w(sw >> 1, bn); else issue_w(c2, sw >> 1);
                        }
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
        if (t + 3 < T) next(c2);
        CV_STAMP(6);
        __syncthreads();
        CV_STAMP(7);
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = wn * 64 + j * 32 + (lane & 31);
            if (col >= nco) continue;
            const float bv = (a.bias && blockIdx.z == 0) ? a.bias[co_blk + col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pix = tile_p + wm * 64 + i * 32 + mfma32_row(r, lane);
                if (pix < L.P) {
                    float v = (acc[i][j][r] + accl[i][j][r]) + bv;
                    if (a.relu) v = fmaxf(v, 0.f);
                    size_t opix = pix;
                    if (a.ostep) {
                        const int HWo = L.Ho * L.Wo;
                        const int b = pix / HWo, rem = pix - b * HWo;
                        const int ho = rem / L.Wo, wo = rem - ho * L.Wo;
                        opix = ((size_t)b * a.OH + a.oy0 + ho * a.ostep) * a.OW + a.ox0 + wo * a.ostep;
                    }
                    float *dst = L.out + opix * a.Co + co_blk + col;
                    if (a.ksplit > 1)
                        atomic_add_f32(dst, v);
                    else
                        *dst = v;
                }
            }
        }
}

// w (Co, kh, kw, C) -> wt (C, kh, kw, Co) with both kernel axes flipped: the weights of the transposed conv
__global__ void conv_flip_transpose_kernel(const float *w, float *wt, int Co, int K, int C)
{
    const size_t n = (size_t)Co * K * C;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(e % Co);
        const size_t r = e / Co;
        const int k = (int)(r % K), ci = (int)(r / K);
        wt[e] = w[((size_t)co * K + (K - 1 - k)) * C + ci];
    }
}

// Tap subset of a transposed convolution: taps i = i0 + m * istep (m < ni), j likewise.
struct TapSub {
    int i0, istep, ni, j0, jstep, nj, kw;
};

// w -> NPL bf16 planes.  flipT = 0: same element order (n = Co*K*C values).  flipT = 1: the source is (Co, K, C) and
// the destination the transposed-conv weight (C, K', Co) over the tap subset `ts` with the taps reversed:
//   dst[ci][m' * nj + n'][co] = w[co][(i0 + (ni-1-m') istep) * kw + j0 + (nj-1-n') jstep][ci]
template <int NPL>
__global__ void conv_prepare_kernel(const float *w, unsigned short *out, int Co, int K, int C, int flipT, TapSub ts)
{
    const int Kd = flipT ? ts.ni * ts.nj : K;
    const size_t n = (size_t)Co * Kd * C;
    for (size_t e = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * 2; e < n; e += (size_t)gridDim.x * blockDim.x * 2) {
        float v[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const size_t d = e + q;
            if (!flipT) {
                v[q] = w[d];
            } else {
                const int co = (int)(d % Co);
                const size_t r = d / Co;
                const int k = (int)(r % Kd), ci = (int)(r / Kd);
                const int m = k / ts.nj, nn = k - m * ts.nj;
                const int i = ts.i0 + (ts.ni - 1 - m) * ts.istep, j = ts.j0 + (ts.nj - 1 - nn) * ts.jstep;
                v[q] = w[((size_t)co * K + i * ts.kw + j) * C + ci];
            }
        }
        unsigned pl[NPL];
        split_planes<NPL>(v[0], v[1], pl);
#pragma unroll
        for (int q = 0; q < NPL; ++q) *reinterpret_cast<unsigned *>(out + q * n + e) = pl[q];
    }
}

void dbg_state(long long **buf, int *block);   // dcn.hip: lsn_debug_phase_clocks
int split_np();   // dcn.hip: bf16 products per fp32 product of the current math mode (0: exact fp32)
static int conv_np() { return split_np() == 3 ? 3 : 6; }   // these kernels have no fp32-MFMA variant: exact mode gets x6

template <int BM, int BN, int NP>
static int launch_conv_np(ConvArgs &a, hipStream_t st)
{
    const size_t lds = (size_t)2 * SplitCfg<NP>::NPL * (BM + BN) * CV_RS;
    int tiles = 0;
    for (int i = 0; i < a.nlv; ++i) {
        a.lv[i].tile0 = tiles;
        tiles += (a.lv[i].P + BM - 1) / BM;
    }
    a.ntiles = tiles;
    const int ncol = (a.Co + BN - 1) / BN;
    // few pixel tiles and a deep reduction (FPN P6 / P7, 1x1 convs on 2048 channels at the smallest maps): split the
    // chunk range over blockIdx.z until the grid covers the chip; partial sums meet in a zero-filled output
    const int Tall = a.kh * a.kw * ((a.C + 31) / 32);
    int ks = 1;
    if (!a.relu && !a.ostep && tiles * ncol < 128 && Tall >= 16) {
        ks = 256 / (tiles * ncol);
        if (ks > Tall / 8) ks = Tall / 8;
        if (ks > 16) ks = 16;
        if (ks < 1) ks = 1;
    }
    a.ksplit = ks;
    dbg_state(&a.dbg, &a.dbg_block);
    if (ks > 1)
        for (int i = 0; i < a.nlv; ++i)
            LSN_HIP(hipMemsetAsync(a.lv[i].out, 0, sizeof(float) * (size_t)a.lv[i].P * a.Co, st));
    dim3 grid(tiles, ncol, ks);
    if (a.wp) {
        auto k = conv_fwd_xn_kernel<BM, BN, true, NP>;
        LSN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k, grid, dim3(256), lds, st, a);
    } else {
        auto k = conv_fwd_xn_kernel<BM, BN, false, NP>;
        LSN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k, grid, dim3(256), lds, st, a);
    }
    LSN_HIP(hipGetLastError());
    return 0;
}

template <int BM, int BN>
static int launch_conv(ConvArgs &a, hipStream_t st)
{
    return conv_np() == 3 ? launch_conv_np<BM, BN, 3>(a, st) : launch_conv_np<BM, BN, 6>(a, st);
}

static void conv_prepare(const float *w, unsigned short *out, int Co, int K, int C, int flipT, const TapSub &ts,
                         hipStream_t st)
{
    const size_t n = (size_t)Co * (flipT ? ts.ni * ts.nj : K) * C;
    const int blocks = (int)((n / 2 + 255) / 256 < 512 ? (n / 2 + 255) / 256 : 512);
    if (conv_np() == 3)
        hipLaunchKernelGGL(conv_prepare_kernel<2>, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, w, out, Co, K, C, flipT, ts);
    else
        hipLaunchKernelGGL(conv_prepare_kernel<3>, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, w, out, Co, K, C, flipT, ts);
}

static int conv_forward(ConvArgs &a, hipStream_t st)
{
    // the block is four 64x64 wave tiles: 1x4 for wide layers, 2x2 for 128 output channels, 4x1 for 64
    static const int force = [] { const char *e = getenv("LSNET_CONV_TILE"); return e ? atoi(e) : 0; }();   // tile sweeps
    if (force == 1) return launch_conv<64, 256>(a, st);
    if (force == 2) return launch_conv<128, 128>(a, st);
    if (force == 3) return launch_conv<256, 64>(a, st);
    if (a.Co <= 64) return launch_conv<256, 64>(a, st);
    if (a.Co <= 128) return launch_conv<128, 128>(a, st);
    return launch_conv<64, 256>(a, st);
}

static int conv_out_size(int in, int k, int stride, int pad, int dil) { return (in + 2 * pad - (dil * (k - 1) + 1)) / stride + 1; }

static int conv_check(int B, int H, int W, int C, int Co, int kh, int kw, int stride, int pad, int dil, int *Ho,
                      int *Wo)
{
    LSN_CHECK(B > 0 && H > 0 && W > 0 && C > 0 && Co > 0 && kh > 0 && kw > 0, "conv2d: empty tensor");
    LSN_CHECK(stride > 0 && dil > 0 && pad >= 0, "conv2d: bad stride / dilation / padding");
    if (C % 4 != 0) return fail(LSN_ERR_UNSUPPORTED, "conv2d kernel needs C %% 4 == 0, got %d", C);
    if (kh * kw > 64) return fail(LSN_ERR_UNSUPPORTED, "conv2d kernel takes at most 64 taps, got %d x %d", kh, kw);
    *Ho = conv_out_size(H, kh, stride, pad, dil);
    *Wo = conv_out_size(W, kw, stride, pad, dil);
    LSN_CHECK(*Ho > 0 && *Wo > 0, "conv2d: output size is too small");
    if ((int64_t)B * H * W * C * 4 >= ((int64_t)1 << 31) || (int64_t)Co * kh * kw * C * 4 >= ((int64_t)1 << 31) ||
        (int64_t)B * *Ho * *Wo * Co * 4 >= ((int64_t)1 << 31))
        return fail(LSN_ERR_UNSUPPORTED, "conv2d: tensor too large for 32-bit buffer offsets");
    return 0;
}

static int conv_forward_impl(int n, const lsn_conv_level *lv, const float *w, const float *bias, void *workspace, int C,
                             int xpitch, int Co, int kh, int kw, int stride, int pad, int dil, int relu, hipStream_t st)
{
    LSN_CHECK(n >= 1 && n <= CV_MAXLV && lv && w, "conv2d: bad level list");
    LSN_CHECK(xpitch > 0 && xpitch % 2 == 0, "conv2d: the pixel pitch must be an even number of floats, got %d", xpitch);
    ConvArgs a = {};
    a.nlv = n;
    for (int i = 0; i < n; ++i) {
        ConvLvl &L = a.lv[i];
        LSN_CHECK(lv[i].x && lv[i].out, "conv2d: NULL tensor in level %d", i);
        if (int rc = conv_check(lv[i].B, lv[i].H, lv[i].W, C, Co, kh, kw, stride, pad, dil, &L.Ho, &L.Wo)) return rc;
        if (xpitch != C) {
            // row-merged form: the C "channels" of a tap are C / xpitch horizontally adjacent pixels; they must stay
            // inside the row (the caller pads the image), and the output grid follows the REAL pixels
            LSN_CHECK(kw == 1 && pad == 0 && dil == 1 && C % xpitch == 0, "conv2d: row-merged form needs kw = 1, pad = 0, dil = 1");
            L.Wo = (lv[i].W - C / xpitch) / stride + 1;
            LSN_CHECK(L.Wo > 0, "conv2d: output size is too small");
        }
        L.x = lv[i].x, L.out = lv[i].out, L.B = lv[i].B, L.H = lv[i].H, L.W = lv[i].W;
        L.P = L.B * L.Ho * L.Wo;
    }
    a.w = w, a.bias = bias;
    if (workspace && C % 8 == 0) {   // split the weights once instead of in every block
        conv_prepare(w, reinterpret_cast<unsigned short *>(workspace), Co, kh * kw, C, 0, TapSub{}, st);
        a.wp = reinterpret_cast<const unsigned short *>(workspace);
    }
    a.C = C, a.Co = Co, a.kh = kh, a.kw = kw, a.stride = stride, a.pad_h = a.pad_w = pad, a.dil = dil;
    a.xpitch = xpitch;
    a.relu = relu;
    return conv_forward(a, st);
}

// grad_in of every level (B, H, W, C) from grad_out (B, Ho, Wo, Co): lv[i].x = grad_out, lv[i].out = grad_in, and
// lv[i].B / H / W are the INPUT sizes of the forward convolution
static int conv_backward_data_impl(int n, const lsn_conv_level *lv, const float *w, float *wt_workspace, int C, int Co,
                                   int kh, int kw, int stride, int pad, int dil, hipStream_t st)
{
    LSN_CHECK(n >= 1 && n <= CV_MAXLV && lv && w && wt_workspace, "conv2d backward: bad arguments");
    if (Co % 8 != 0) return fail(LSN_ERR_UNSUPPORTED, "conv2d backward-data kernel needs Co %% 8 == 0");
    const int K = kh * kw, s = stride;
    int Ho[CV_MAXLV], Wo[CV_MAXLV];
    for (int i = 0; i < n; ++i) {
        LSN_CHECK(lv[i].x && lv[i].out, "conv2d backward: NULL tensor in level %d", i);
        if (int rc = conv_check(lv[i].B, lv[i].H, lv[i].W, C, Co, kh, kw, stride, pad, dil, &Ho[i], &Wo[i])) return rc;
    }
    if (s > 1 && n > 1) return fail(LSN_ERR_UNSUPPORTED, "conv2d backward-data: strided convolutions take one level per call");
    // grad_in[y, x] = sum over taps (i, j) with (y + pad - i dil) % s == 0 (same for x) of
    //                 grad_out[(y + pad - i dil) / s, (x + pad - j dil) / s] . w[:, i, j, :]
    // Each residue class (y % s, x % s) of input pixels is a stride-1 convolution of grad_out with its own subset of
    // the taps (an arithmetic progression): no zero-stuffed samples, no wasted products.  Classes without a tap keep
    // the zeros of the memset.
    if (s * s > 64) return fail(LSN_ERR_UNSUPPORTED, "conv2d backward-data: stride %d is not supported", s);
    const int H = lv[0].H, W = lv[0].W, B = lv[0].B;
    bool need_zero = false;
    struct Cls {
        TapSub ts;
        int py, px, Hc, Wc, pad_h, pad_w, dstep;
    } cls[64];
    int ncls = 0;
    auto taps_of = [&](int p, int k, int &t0, int &tstep, int &nt) {
        t0 = -1, tstep = 1, nt = 0;
        int prev = -1;
        for (int t = 0; t < k; ++t)
            if (((p + pad - t * dil) % s + s) % s == 0) {
                if (nt == 0) t0 = t;
                if (nt == 1) tstep = t - prev;
                prev = t;
                ++nt;
            }
    };
    for (int py = 0; py < s; ++py)
        for (int px = 0; px < s; ++px) {
            Cls c;
            c.py = py, c.px = px;
            c.Hc = py < H ? (H - py + s - 1) / s : 0;
            c.Wc = px < W ? (W - px + s - 1) / s : 0;
            if (s > 1 && (c.Hc == 0 || c.Wc == 0)) continue;
            taps_of(py, kh, c.ts.i0, c.ts.istep, c.ts.ni);
            taps_of(px, kw, c.ts.j0, c.ts.jstep, c.ts.nj);
            c.ts.kw = kw;
            if (c.ts.ni == 0 || c.ts.nj == 0) {
                need_zero = true;
                continue;
            }
            // y_in = yy + q0 - m d' (m-th tap of the subset), d' = istep dil / s; as a correlation over the reversed
            // subset: y_in = yy - P + m' d' with P = (ni - 1) d' - q0
            const int dstep_h = c.ts.istep * dil / s, dstep_w = c.ts.jstep * dil / s;
            if (c.ts.ni > 1 && c.ts.nj > 1 && dstep_h != dstep_w)
                return fail(LSN_ERR_UNSUPPORTED, "conv2d backward-data: unequal tap spacing");
            c.dstep = c.ts.ni > 1 ? dstep_h : (c.ts.nj > 1 ? dstep_w : 1);
            if ((c.ts.ni > 1 && c.ts.istep * dil % s != 0) || (c.ts.nj > 1 && c.ts.jstep * dil % s != 0))
                return fail(LSN_ERR_UNSUPPORTED, "conv2d backward-data: stride / dilation combination");
            const int q0h = (py + pad - c.ts.i0 * dil) / s, q0w = (px + pad - c.ts.j0 * dil) / s;
            c.pad_h = (c.ts.ni - 1) * c.dstep - q0h;
            c.pad_w = (c.ts.nj - 1) * c.dstep - q0w;
            cls[ncls++] = c;
        }
    if (need_zero) LSN_HIP(hipMemsetAsync(lv[0].out, 0, sizeof(float) * (size_t)B * H * W * C, st));
    size_t ws_off = 0;   // every class gets its own slice of the workspace (the launches are asynchronous)
    for (int ci = 0; ci < ncls; ++ci) {
        const Cls &c = cls[ci];
        const int Kc = c.ts.ni * c.ts.nj;
        unsigned short *wp = reinterpret_cast<unsigned short *>(wt_workspace) + ws_off;
        conv_prepare(w, wp, Co, K, C, 1, c.ts, st);
        ws_off += (size_t)3 * Co * Kc * C;
        ConvArgs a = {};
        a.wp = wp;
        a.nlv = n;
        for (int i = 0; i < n; ++i) {
            ConvLvl &L = a.lv[i];
            L.x = lv[i].x, L.out = lv[i].out;
            L.B = lv[i].B, L.H = Ho[i], L.W = Wo[i];
            L.Ho = s > 1 ? c.Hc : lv[i].H, L.Wo = s > 1 ? c.Wc : lv[i].W;
            L.P = L.B * L.Ho * L.Wo;
        }
        a.C = Co, a.Co = C, a.kh = c.ts.ni, a.kw = c.ts.nj, a.stride = 1;
        a.xpitch = Co;
        a.pad_h = c.pad_h, a.pad_w = c.pad_w, a.dil = c.dstep;
        if (s > 1) a.ostep = s, a.oy0 = c.py, a.ox0 = c.px, a.OH = H, a.OW = W;
        if (int rc = conv_forward(a, st)) return rc;
    }
    return 0;
}

}  // namespace lsn

extern "C" {

int lsn_conv2d_forward_multi(int n_levels, const lsn_conv_level *levels, const float *w, const float *bias, void *workspace,
                             int C, int Co, int kh, int kw, int stride, int pad, int dil, int relu, lsn_stream_t stream)
{
    return lsn::conv_forward_impl(n_levels, levels, w, bias, workspace, C, C, Co, kh, kw, stride, pad, dil, relu,
                                  reinterpret_cast<hipStream_t>(stream));
}

int lsn_conv2d_backward_data_multi(int n_levels, const lsn_conv_level *levels, const float *w, float *wt_workspace, int C,
                                   int Co, int kh, int kw, int stride, int pad, int dil, lsn_stream_t stream)
{
    return lsn::conv_backward_data_impl(n_levels, levels, w, wt_workspace, C, Co, kh, kw, stride, pad, dil,
                                        reinterpret_cast<hipStream_t>(stream));
}

int lsn_conv2d_forward_pitched(const float *x, const float *w, const float *bias, float *out, void *workspace, int B,
                               int H, int W, int C, int xpitch, int Co, int kh, int kw, int stride, int pad, int dil,
                               int relu, lsn_stream_t stream)
{
    lsn_conv_level L = {};
    L.x = x, L.out = out, L.B = B, L.H = H, L.W = W;
    return lsn::conv_forward_impl(1, &L, w, bias, workspace, C, xpitch, Co, kh, kw, stride, pad, dil, relu,
                                  reinterpret_cast<hipStream_t>(stream));
}

int lsn_conv2d_forward(const float *x, const float *w, const float *bias, float *out, void *workspace, int B, int H,
                       int W, int C, int Co, int kh, int kw, int stride, int pad, int dil, int relu,
                       lsn_stream_t stream)
{
    return lsn_conv2d_forward_pitched(x, w, bias, out, workspace, B, H, W, C, C, Co, kh, kw, stride, pad, dil, relu, stream);
}

int lsn_conv2d_backward_data(const float *grad_out, const float *w, float *grad_in, float *wt_workspace, int B,
                             int H, int W, int C, int Co, int kh, int kw, int stride, int pad, int dil,
                             lsn_stream_t stream)
{
    lsn_conv_level L = {};
    L.x = grad_out, L.out = grad_in, L.B = B, L.H = H, L.W = W;
    return lsn::conv_backward_data_impl(1, &L, w, wt_workspace, C, Co, kh, kw, stride, pad, dil,
                                        reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
