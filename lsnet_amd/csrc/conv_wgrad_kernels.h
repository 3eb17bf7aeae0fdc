// Weight / bias gradient of a dense convolution (included by conv.hip):
//
//   gw[co][tap][ci] = sum_p go[p][co] * x[p @ tap][ci],      gb[co] = sum_p go[p][co]
//
// an (co x (tap, ci)) GEMM whose reduction index is the PIXEL, on channels-last fp32 tensors -- both operands arrive
// with the reduction index as their slow dimension.  On gfx950 that needs no software transpose: the LDS images keep
// the natural [pixel][channel] rows (split into bf16 planes while staged) and the MFMA fragments are read with
// ds_read_b64_tr_b16, which hands a lane four consecutive pixels of its own channel (tools/ubench/tr16_probe.hip pins
// the semantics: within a 16-lane group lane i supplies the address of 8-byte chunk i, chunks 4r .. 4r+3 are row r, and
// lane i receives [row 0..3][column i]).
//
// One k-step = a segment of 16 output pixels of one image row.  For it the workgroup stages ONCE
//   * the grad_output rows: 16 pixels x BM output channels, and
//   * the input PATCH the segment's taps touch: kh rows x (15 stride + (kw-1) dil + 1) pixels x BN input channels --
//     3 x 18 pixels serve all nine taps of a 3x3 convolution, 2.7 x fewer values to fetch and split than nine shifted
//     16-pixel tiles, and every one of them is split exactly once per (co, ci) block instead of once per tap;
// every tap's A fragment is then a tr-read of the same patch at a shifted pixel.  A wave owns a 32 ci x 32 co tile for
// all TG taps (TG = 9: nine accumulator tiles) or 2 x 2 tiles for one tap (1x1 convolutions); the workgroup 2 x 2 (or,
// for the 27-channel offset convolutions, 1 x 4) of those.  32-byte slots of the 128 / 256-byte LDS rows are XOR-swizzled
// so that the four pixel rows x two column groups of a half-wave's tr-read fall on different banks.
//
// The pixel range is split over blockIdx.y; every split stores its partial tile with plain 16-byte stores and
// conv_wgrad_reduce_kernel adds the splits (into the gradient, or -- accumulate -- onto it).  fp32 atomics from 512
// workgroups into a 2 MB gradient were measured slower than the whole contraction (profiles/r3_conv_ksplit.txt).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "conv_kernels.h"

namespace lsn {

struct WgLvl {
    const float *x, *go;
    int B, H, W, Ho, Wo;
    int nsx;    // segments per output row: ceil(Wo / 16)
    int seg0;   // first segment of this level
};

struct WgArgs {
    WgLvl lv[CV_MAXLV];
    int nlv, nseg;
    int C, Co, kh, kw, stride, pad, dil;
    int PW;        // patch width in stored pixels; cstep: input pixels between stored patch columns (1x1 strided: stride)
    int cstep;
    float *part;   // [split][Co * K * C] partial gradients, then [split][Co] partial bias gradients at part_b
    float *part_b;
    int want_bias;
};

// TI x TJ tiles of 32 ci x 32 co per wave and tap, TG taps (9: a 3x3 kernel, 1: one tap), waves WI (ci) x WJ (co).
// GUNAL: Co % 4 != 0 (27-channel offset / mask convolutions): grad_output rows are fetched with guarded 4-byte loads.
// PMAX: patch pixels the instantiation stages, rounded up to whole passes of the 256 threads (54 = 3 x 18: stride-1
// 3x3 with dilation 1; 16: one tap) -- every thread always fetches, splits and stores its slots: no exec-masked branches.
// FINE (experiment, LSNET_WGRAD_FINE=1): as in conv_mm_kernel -- slice commits without their `t + 1 < T` branch (past the last
// segment the loads are out-of-bounds zeros: harmless, also for the bias sum) + sched_group_barrier groups, so that the
// split instructions of a slice sit between the MFMAs instead of behind them.
template <int TI, int TJ, int TG, int WI, int WJ, int NP, bool GUNAL, int PMAX, bool FINE = false>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgArgs a)
{
    using SC = SplitCfg<NP>;
    constexpr int NPL = SC::NPL;
    static_assert(WI * WJ == 4, "four waves");
    static_assert(TG == 1 || TG == 9, "one tap or a 3x3 kernel");
    constexpr int KW = TG == 9 ? 3 : 1;
    constexpr int BN = WI * TI * 32;          // input channels of the block
    constexpr int BM = WJ * TJ * 32;          // output channels of the block
    constexpr int RBX = BN * 2, RBG = BM * 2;  // LDS row bytes (bf16)
    constexpr int XT = BN / 4, XP = 256 / XT;   // patch: threads per pixel, pixels per pass
    constexpr int XL = (PMAX + XP - 1) / XP;    // float4 loads per thread
    constexpr int NPIX = XL * XP;               // LDS rows of the patch image (>= the patch: padded to whole passes)
    constexpr int GT = BM / 4, GP = 256 / GT;
    constexpr int GL = (16 + GP - 1) / GP;
    constexpr int NGPX = GL * GP;
    constexpr int XPL = NPIX * RBX, GPL = NGPX * RBG, STAGE = NPL * (XPL + GPL);
    constexpr int NSL = XL + GL;                // staging slices of a k-step
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __align__(16) unsigned char smem[];   // 2 x STAGE: [x planes][go planes]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave / WJ, wj = wave % WJ;
    const int npix = (TG == 9 ? 3 : 1) * a.PW;
    const int nib = (a.C + BN - 1) / BN;
    const int work = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int split = work / (int)gridDim.x, blk = work - split * (int)gridDim.x;
    const int cb = blk / nib, ib = blk - cb * nib;
    const int ci_blk = ib * BN, co_blk = cb * BM;
    // (32-bit quotients: the host checks nseg * splits < 2^32; 64-bit ones are two ~300-instruction scalar software divisions)
    const int s_begin = (int)((unsigned)a.nseg * (unsigned)split / gridDim.y);
    const int T = (int)((unsigned)a.nseg * (unsigned)(split + 1) / gridDim.y) - s_begin;

    // ---- staging roles (constant over segments) ----
    const int xc4 = tid % XT;
    int xrow[XL], xcol[XL], xlds[XL];
    bool xin[XL];
#pragma unroll
    for (int it = 0; it < XL; ++it) {
        const int q = it * XP + tid / XT;
        xin[it] = q < npix && ci_blk + 4 * xc4 < a.C;
        const int r = q / a.PW;
        xrow[it] = r, xcol[it] = q - r * a.PW;
        // 32-byte slot swizzle: slot ^ 2 * ((pixel >> (RBX == 128)) & (RBX / 64 - 1))
        const int sl = (xc4 >> 2) ^ (2 * ((q >> (RBX == 128 ? 1 : 0)) & (RBX / 64 - 1)));
        xlds[it] = q * RBX + sl * 32 + (xc4 & 3) * 8;
    }
    const int gc4 = tid % GT;
    int glds[GL];
    bool gin[GL];
#pragma unroll
    for (int it = 0; it < GL; ++it) {
        const int q = it * GP + tid / GT;
        gin[it] = q < 16 && co_blk + 4 * gc4 < a.Co;
        const int sl = (gc4 >> 2) ^ (2 * ((q >> (RBG == 128 ? 1 : 0)) & (RBG / 64 - 1)));
        glds[it] = q * RBG + sl * 32 + (gc4 & 3) * 8;
    }

    float4 xv[XL], gv[GL];
    float bacc[4] = {0.f, 0.f, 0.f, 0.f};
    const bool do_bias = a.want_bias && ib == 0;

    // segment under issue: level, image / row / first column, buffer descriptors (scalars)
    struct Seg {
        int b, ho, wo0, y0, x0, H, W, Ho, Wo;
        __amdgpu_buffer_rsrc_t xrs, grs;
        bool live;
        int s, sx, nsx, seg_next;   // this segment, its place in the output row, first segment of the next level
    } sg;
    auto open_seg = [&](int s, bool live) {
        int li = 0;
        while (li + 1 < a.nlv && s >= a.lv[li + 1].seg0) ++li;
        const WgLvl &L = a.lv[li];
        const int r = s - L.seg0;
        const int rowid = r / L.nsx, sx = r - rowid * L.nsx;
        sg.b = rowid / L.Ho, sg.ho = rowid - sg.b * L.Ho;
        sg.wo0 = sx * 16;
        sg.y0 = sg.ho * a.stride - a.pad, sg.x0 = sg.wo0 * a.stride - a.pad;
        sg.H = L.H, sg.W = L.W, sg.Ho = L.Ho, sg.Wo = L.Wo;
        sg.xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(L.x), 0, L.B * L.H * L.W * a.C * 4, 0x00020000);
        sg.grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(L.go), 0, L.B * L.Ho * L.Wo * a.Co * 4, 0x00020000);
        sg.live = live;
        sg.s = s, sg.sx = sx, sg.nsx = L.nsx;
        sg.seg_next = li + 1 < a.nlv ? a.lv[li + 1].seg0 : 0x7fffffff;
    };
    // Round 6: the segment BEHIND the open one by stepping -- the next 16 pixels of the row, the next row, the next image; the
    // level search, the two integer divisions and the descriptors of open_seg (a dependent scalar chain of ~500 cycles at the
    // top of EVERY k-step, which is 768 cycles of MFMAs for the 2 x 2 tile) only where the level changes.
    auto next_seg = [&](bool live) {
        const int s = sg.s + 1;
        if (s >= sg.seg_next) {
            open_seg(s, live);
            return;
        }
        sg.s = s;
        if (++sg.sx == sg.nsx) {
            sg.sx = 0;
            if (++sg.ho == sg.Ho) sg.ho = 0, ++sg.b;
        }
        sg.wo0 = sg.sx * 16;
        sg.y0 = sg.ho * a.stride - a.pad, sg.x0 = sg.wo0 * a.stride - a.pad;
        sg.live = live;
    };
    auto issue_slice = [&](int sl) {   // slices 0 .. XL-1: patch passes; XL .. NSL-1: grad_output passes
        if (sl < XL) {
            const int it = sl;
            const int y = sg.y0 + xrow[it] * a.dil, x = sg.x0 + xcol[it] * a.cstep;
            const bool ok = sg.live && xin[it] && (unsigned)y < (unsigned)sg.H && (unsigned)x < (unsigned)sg.W;
            xv[it] = cv_load4(sg.xrs, ok ? (((sg.b * sg.H + y) * sg.W + x) * a.C + ci_blk + 4 * xc4) * 4 : OOB, 0);
        } else {
            const int it = sl - XL;
            const int wo = sg.wo0 + it * GP + tid / GT;
            const bool ok = sg.live && gin[it] && wo < sg.Wo;
            const int vo = (((sg.b * sg.Ho + sg.ho) * sg.Wo + wo) * a.Co + co_blk + 4 * gc4) * 4;
            if constexpr (!GUNAL) {
                gv[it] = cv_load4(sg.grs, ok ? vo : OOB, 0);
            } else {
                const int c0 = co_blk + 4 * gc4;
                gv[it].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sg.grs, ok ? vo : OOB, 0, 0));
                gv[it].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sg.grs, ok && c0 + 1 < a.Co ? vo + 4 : OOB, 0, 0));
                gv[it].z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sg.grs, ok && c0 + 2 < a.Co ? vo + 8 : OOB, 0, 0));
                gv[it].w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sg.grs, ok && c0 + 3 < a.Co ? vo + 12 : OOB, 0, 0));
            }
        }
    };
    auto commit_slice = [&](int sl, unsigned char *buf) {   // split the fetched values into planes -> LDS stage `buf`
        unsigned p0[NPL], p1[NPL];
        if (sl < XL) {
            const int it = sl;
            split_planes<NPL>(xv[it].x, xv[it].y, p0);
            split_planes<NPL>(xv[it].z, xv[it].w, p1);
#pragma unroll
            for (int q = 0; q < NPL; ++q) *reinterpret_cast<uint2 *>(buf + q * XPL + xlds[it]) = make_uint2(p0[q], p1[q]);
        } else {
            const int it = sl - XL;
            if constexpr (FINE) {   // branch-free: a branch would end the basic block the scheduler interleaves in
                const float bm = do_bias ? 1.f : 0.f;
                bacc[0] += bm * gv[it].x, bacc[1] += bm * gv[it].y, bacc[2] += bm * gv[it].z, bacc[3] += bm * gv[it].w;
            } else if (do_bias) {
                bacc[0] += gv[it].x, bacc[1] += gv[it].y, bacc[2] += gv[it].z, bacc[3] += gv[it].w;
            }
            split_planes<NPL>(gv[it].x, gv[it].y, p0);
            split_planes<NPL>(gv[it].z, gv[it].w, p1);
#pragma unroll
            for (int q = 0; q < NPL; ++q)
                *reinterpret_cast<uint2 *>(buf + NPL * XPL + q * GPL + glds[it]) = make_uint2(p0[q], p1[q]);
        }
    };

    // ---- fragment addresses (constant over segments) ----
    // lane: column group g = (lane >> 4) & 1 of the 32-wide tile, chunk i16 = lane & 15: block row rb = i16 >> 2, 8-byte
    // piece cq = i16 & 3; pixels m = 8 (lane >> 5) + 4 h + rb for the two tr-reads h = 0, 1 of a k-step of 16 pixels
    const int g = (lane >> 4) & 1, rb = (lane >> 2) & 3, cq = lane & 3;
    int xaddr[TG][TI][2];   // patch: per tap, ci tile, h
#pragma unroll
    for (int tp = 0; tp < TG; ++tp)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int m = 8 * (lane >> 5) + 4 * h + rb;
                const int pp = (tp / KW) * a.PW + (m * a.stride + (tp % KW) * a.dil) / a.cstep;
                const int sl = (2 * (wi * TI + i) + g) ^ (2 * ((pp >> (RBX == 128 ? 1 : 0)) & (RBX / 64 - 1)));
                xaddr[tp][i][h] = pp * RBX + sl * 32 + cq * 8;
            }
    int gaddr[TJ][2];
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = 8 * (lane >> 5) + 4 * h + rb;
            const int sl = (2 * (wj * TJ + j) + g) ^ (2 * ((m >> (RBG == 128 ? 1 : 0)) & (RBG / 64 - 1)));
            gaddr[j][h] = m * RBG + sl * 32 + cq * 8;
        }

    f32x16 acc[TG][TI][TJ];
#pragma unroll
    for (int tp = 0; tp < TG; ++tp)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tp][i][j][r] = 0.f;

    auto frag = [&](const unsigned char *p0, const unsigned char *p1) {
        // (the two 8-byte reads side by side as whole dwords.  Assembled from eight 16-bit elements hipcc emitted a v_bfi_b32 per
        // dword -- arithmetically a no-op, but it CONSUMES the read: every fragment was waited for where it was requested, at
        // the top of the iteration, instead of in front of the MFMA that takes it half an iteration later)
        const s16x4 lo = lds_tr16(p0), hi = lds_tr16(p1);
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x2 a2 = __builtin_bit_cast(u32x2, lo), b2 = __builtin_bit_cast(u32x2, hi);
        const u32x4 r = {a2.x, a2.y, b2.x, b2.y};
        return __builtin_bit_cast(bf16x8, r);
    };

    // ---- prologue: segment 0 -> stage 0, loads of segment 1 in flight ----
    open_seg(s_begin, T > 0);
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) issue_slice(sl);
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) commit_slice(sl, smem);
    next_seg(T > 1);
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) issue_slice(sl);
    __syncthreads();

    // One iteration = one k-step (16 pixels).  Tap tp's MFMAs are issued behind the fragment reads of tap tp + 1, and one
    // staging slice (split of segment t + 1 -> the other LDS stage, then the load of segment t + 2 into the freed
    // registers) rides behind each of the first NSL taps; the fences keep that order.
    for (int t = 0; t < T; ++t) {
        const unsigned char *bc = smem + (t & 1) * STAGE;
        unsigned char *bn = smem + ((t & 1) ^ 1) * STAGE;
        bf16x8 Gf[TJ][NPL], Xf[2][TI][NPL];
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int q = 0; q < NPL; ++q)
                Gf[j][q] = frag(bc + NPL * XPL + q * GPL + gaddr[j][0], bc + NPL * XPL + q * GPL + gaddr[j][1]);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int q = 0; q < NPL; ++q) Xf[0][i][q] = frag(bc + q * XPL + xaddr[0][i][0], bc + q * XPL + xaddr[0][i][1]);
        next_seg(t + 2 < T);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (FINE && TG == 1) {   // one tap: its MFMAs in NSL parts, one staging slice behind each (conv_mm_kernel)
            constexpr int NMt = NP * TI * TJ;
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl) {
#pragma unroll
                for (int m = sl * NMt / NSL; m < (sl + 1) * NMt / NSL; ++m) {
                    const int prod = m / (TI * TJ), i = (m / TJ) % TI, j = m % TJ;
                    acc[0][i][j] = mfma_bf16(Xf[0][i][SC::pa(prod)], Gf[j][SC::pb(prod)], acc[0][i][j]);
                }
                commit_slice(sl, bn);
                issue_slice(sl);
#pragma unroll
                for (int g = 0; g < NMt / NSL; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);   // vector / scalar ALU instructions
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else
#pragma unroll
        for (int tp = 0; tp < TG; ++tp) {
            if (tp + 1 < TG) {
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int q = 0; q < NPL; ++q)
                        Xf[(tp + 1) & 1][i][q] = frag(bc + q * XPL + xaddr[tp + 1][i][0], bc + q * XPL + xaddr[tp + 1][i][1]);
            }
#pragma unroll
            for (int prod = 0; prod < NP; ++prod)
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[tp][i][j] = mfma_bf16(Xf[tp & 1][i][SC::pa(prod)], Gf[j][SC::pb(prod)], acc[tp][i][j]);
            // staging slices: TG == 9 spreads them over the taps, a single tap carries them all
#pragma unroll
            for (int sl = (TG == 1 ? 0 : tp); sl < (TG == 1 ? NSL : (tp < NSL ? tp + 1 : 0)); ++sl) {
                if (FINE || t + 1 < T) commit_slice(sl, bn);
                issue_slice(sl);
            }
            if (TG == 9 && tp == TG - 1) {   // slices that did not fit behind a tap (never with 5 slices and 9 taps)
#pragma unroll
                for (int sl = TG; sl < NSL; ++sl) {
                    if (FINE || t + 1 < T) commit_slice(sl, bn);
                    issue_slice(sl);
                }
            }
            if constexpr (FINE) {
#pragma unroll
                for (int g = 0; g < NP * TI * TJ; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);   // vector / scalar ALU instructions
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // ---- epilogue: D[ci][co]: lane = co (lane & 31) of tile j, input channels 8 q + 4 (lane >> 5) + (0..3) of tile i ----
    const size_t nW = (size_t)a.Co * TG * a.C;
    float *pw = a.part + (size_t)split * nW;
    const bool v4 = (a.C & 3) == 0;
#pragma unroll
    for (int tp = 0; tp < TG; ++tp) {
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int co = co_blk + (wj * TJ + j) * 32 + (lane & 31);
            if (co >= a.Co) continue;
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ci = ci_blk + (wi * TI + i) * 32 + 8 * q + 4 * (lane >> 5);
                    if (ci >= a.C) continue;
                    float *dst = pw + ((size_t)co * TG + tp) * a.C + ci;
                    if (v4) {
                        *reinterpret_cast<float4 *>(dst) = make_float4(acc[tp][i][j][4 * q], acc[tp][i][j][4 * q + 1],
                                                                       acc[tp][i][j][4 * q + 2], acc[tp][i][j][4 * q + 3]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (ci + e < a.C) dst[e] = acc[tp][i][j][4 * q + e];
                    }
                }
        }
    }
    if (do_bias) {   // 256 / GT threads hold partial sums of the same four output channels: meet in LDS
        __syncthreads();
        float *red = reinterpret_cast<float *>(smem);
#pragma unroll
        for (int e = 0; e < 4; ++e) red[(tid / GT) * BM + 4 * gc4 + e] = bacc[e];
        __syncthreads();
        if (tid < BM && co_blk + tid < a.Co) {
            float s = 0.f;
            for (int r = 0; r < GP; ++r) s += red[r * BM + tid];
            a.part_b[(size_t)split * a.Co + co_blk + tid] = s;
        }
    }
}

// gw[e] (+)= sum_s part[s][e] over n elements (n % 4 == 0); the bias part likewise (nb elements) -- one launch for
// both (the bias partials may come in their own number, splits_b).  LS (a power of two <= 64) adjacent lanes share one float4 of the gradient and take every LS-th split each, then
// meet by xor-shuffles: a small gradient under many splits (64 x 64 weights, 512 pixel ranges) is then summed by
// 64 lanes x 8 loads instead of one thread walking 512 dependent loads.
// (vblock of vgrid: the block's place in ITS job's share of the grid -- the single-job kernel passes blockIdx / gridDim, the
// multi-job kernel of a deferred flush the position inside the job's block range)
__device__ __forceinline__ void wgrad_reduce_body(const float *__restrict__ part, float *__restrict__ gw, size_t n,
                                                  const float *__restrict__ part_b, float *__restrict__ gb, int nb, int splits,
                                                  int splits_b, int accumulate, int LS, unsigned vblock, unsigned vgrid)
{
    const size_t n4 = n / 4;
    const size_t gid = vblock * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)vgrid * blockDim.x / LS;
    const int sl = (int)(gid % LS);
    for (size_t e0 = gid / LS; e0 < ((n4 + stride - 1) / stride) * stride; e0 += stride) {   // (whole waves stay in the loop)
        const bool live = e0 < n4;
        const size_t e = live ? e0 : 0;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int z = sl;
        // four partial tiles in flight per lane (the summation order stays z, z + LS, ...: the same bits as a rolled loop,
        // whose every addition waited for its own load -- 8 dependent memory latencies for 32 splits under LS = 4)
        for (; z + 3 * LS < splits; z += 4 * LS) {
            const float4 p0 = reinterpret_cast<const float4 *>(part + (size_t)z * n)[e];
            const float4 p1 = reinterpret_cast<const float4 *>(part + (size_t)(z + LS) * n)[e];
            const float4 p2 = reinterpret_cast<const float4 *>(part + (size_t)(z + 2 * LS) * n)[e];
            const float4 p3 = reinterpret_cast<const float4 *>(part + (size_t)(z + 3 * LS) * n)[e];
            s.x += p0.x, s.y += p0.y, s.z += p0.z, s.w += p0.w;
            s.x += p1.x, s.y += p1.y, s.z += p1.z, s.w += p1.w;
            s.x += p2.x, s.y += p2.y, s.z += p2.z, s.w += p2.w;
            s.x += p3.x, s.y += p3.y, s.z += p3.z, s.w += p3.w;
        }
        for (; z < splits; z += LS) {
            const float4 p = reinterpret_cast<const float4 *>(part + (size_t)z * n)[e];
            s.x += p.x, s.y += p.y, s.z += p.z, s.w += p.w;
        }
        for (int m = 1; m < LS; m <<= 1) {
            s.x += __shfl_xor(s.x, m), s.y += __shfl_xor(s.y, m), s.z += __shfl_xor(s.z, m), s.w += __shfl_xor(s.w, m);
        }
        if (live && sl == 0) {
            float4 *d = reinterpret_cast<float4 *>(gw) + e;
            if (accumulate) {
                const float4 o = *d;
                s.x += o.x, s.y += o.y, s.z += o.z, s.w += o.w;
            }
            *d = s;
        }
    }
    if (gb && vblock == 0)
        for (int e = threadIdx.x; e < nb; e += blockDim.x) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int z = 0;
            for (; z + 3 < splits_b; z += 4) {
                s0 += part_b[(size_t)z * nb + e], s1 += part_b[(size_t)(z + 1) * nb + e];
                s2 += part_b[(size_t)(z + 2) * nb + e], s3 += part_b[(size_t)(z + 3) * nb + e];
            }
            for (; z < splits_b; ++z) s0 += part_b[(size_t)z * nb + e];
            const float s = (s0 + s1) + (s2 + s3);
            gb[e] = accumulate ? gb[e] + s : s;
        }
}

static __global__ void conv_wgrad_reduce_kernel(const float *__restrict__ part, float *__restrict__ gw, size_t n,
                                                const float *__restrict__ part_b, float *__restrict__ gb, int nb, int splits,
                                                int splits_b, int accumulate, int LS)
{
    wgrad_reduce_body(part, gw, n, part_b, gb, nb, splits, splits_b, accumulate, LS, blockIdx.x, gridDim.x);
}

// The reduce of a weight gradient whose convolution carries a folded eval-mode BatchNorm (lsn_conv2d_backward_weight_bn):
// part holds the partial gradients G' of the RAW convolution under the gated upstream gradient g, part_b the partial sums
// of g per output channel.  One workgroup per output channel co (row of R = K C floats, R % 4 == 0):
//   G[co][:]       = sum_s part[s][co][:]
//   gw[co][:]      (+)= a_co G[co][:],                 a_co = gamma_co / sqrt(var_co + eps)
//   dbeta[co]      (+)= sum_s part_b[s][co]
//   dgamma[co]     (+)= (w[co][:] . G[co][:] - mean_co dbeta_co) / sqrt(var_co + eps)
// LS adjacent lanes share one float4 and take every LS-th split each (as in conv_wgrad_reduce_kernel); the dot product
// and the channel sum meet by xor-shuffles in a fixed order: the same bits on every run.
struct WgFold {
    const float *w, *gamma, *mean, *var;
    float *dgamma, *dbeta;
    float eps;
};
// Several convolutions of ONE geometry in one launch (lsn_conv2d_backward_weight_bn_jobs: the identical bottlenecks of a
// ResNet stage): job j is level j of the main kernel, the pixel splits [j S, (j + 1) S) are its partial tiles, and
// blockIdx.y of the reduce is the job.
struct WgFoldJobs {
    WgFold f[8];
    float *gw[8];
    int njobs;
};

// one output channel of one job (the single-job kernel: blockIdx.x of job blockIdx.y; the multi-job kernel of a deferred flush:
// the channel inside the job's block range)
__device__ __forceinline__ void wgrad_reduce_bn_body(const float *__restrict__ part, float *gw, int R, int Co,
                                                     const float *__restrict__ part_b, int splits, int splits_b, int accumulate,
                                                     int LS, const WgFold &f, int co)
{
    // one workgroup per output channel (a first version with one WAVE per channel ran 32 .. 512 waves through up to 32
    // dependent passes each: +1.9 ms per step over the plain reduce, profiles/r4_bench_c03.log)
    __shared__ float red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t n = (size_t)Co * R;
    const int sl = tid % LS, el = tid / LS, EP = 256 / LS;   // split lane, element lane, elements per pass (LS | 64)
    const float rstd = rsqrtf(f.var[co] + f.eps), ac = f.gamma[co] * rstd;
    const int R4 = R / 4;
    float dot = 0.f;
    for (int e0 = 0; e0 < R4; e0 += EP) {
        const bool live = e0 + el < R4;
        const size_t e = (size_t)co * R4 + (live ? e0 + el : 0);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int z = sl;
        // four partial tiles in flight per lane (the summation order stays z, z + LS, ...: the same bits as a rolled loop,
        // whose every addition waited for its own load -- 8 dependent memory latencies for 32 splits under LS = 4)
        for (; z + 3 * LS < splits; z += 4 * LS) {
            const float4 p0 = reinterpret_cast<const float4 *>(part + (size_t)z * n)[e];
            const float4 p1 = reinterpret_cast<const float4 *>(part + (size_t)(z + LS) * n)[e];
            const float4 p2 = reinterpret_cast<const float4 *>(part + (size_t)(z + 2 * LS) * n)[e];
            const float4 p3 = reinterpret_cast<const float4 *>(part + (size_t)(z + 3 * LS) * n)[e];
            s.x += p0.x, s.y += p0.y, s.z += p0.z, s.w += p0.w;
            s.x += p1.x, s.y += p1.y, s.z += p1.z, s.w += p1.w;
            s.x += p2.x, s.y += p2.y, s.z += p2.z, s.w += p2.w;
            s.x += p3.x, s.y += p3.y, s.z += p3.z, s.w += p3.w;
        }
        for (; z < splits; z += LS) {
            const float4 p = reinterpret_cast<const float4 *>(part + (size_t)z * n)[e];
            s.x += p.x, s.y += p.y, s.z += p.z, s.w += p.w;
        }
        for (int m = 1; m < LS; m <<= 1) {
            s.x += __shfl_xor(s.x, m), s.y += __shfl_xor(s.y, m), s.z += __shfl_xor(s.z, m), s.w += __shfl_xor(s.w, m);
        }
        if (live && sl == 0) {
            const float4 wv = reinterpret_cast<const float4 *>(f.w)[e];
            dot += (wv.x * s.x + wv.y * s.y) + (wv.z * s.z + wv.w * s.w);
            float4 *d = reinterpret_cast<float4 *>(gw) + e;
            float4 o = make_float4(ac * s.x, ac * s.y, ac * s.z, ac * s.w);
            if (accumulate) {
                const float4 q = *d;
                o.x += q.x, o.y += q.y, o.z += q.z, o.w += q.w;
            }
            *d = o;
        }
    }
    float sb = 0.f;
    for (int z = tid; z < splits_b; z += 256) sb += part_b[(size_t)z * Co + co];
    for (int m = 1; m < 64; m <<= 1) dot += __shfl_xor(dot, m), sb += __shfl_xor(sb, m);
    if (lane == 0) red[0][wave] = dot, red[1][wave] = sb;
    __syncthreads();
    if (tid == 0) {
        const float d4 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        const float b4 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        const float dg = (d4 - f.mean[co] * b4) * rstd;
        f.dbeta[co] = accumulate ? f.dbeta[co] + b4 : b4;
        f.dgamma[co] = accumulate ? f.dgamma[co] + dg : dg;
    }
}

static __global__ __launch_bounds__(256) void conv_wgrad_reduce_bn_kernel(const float *__restrict__ part, float *gw, int R, int Co,
                                                                           const float *__restrict__ part_b, int splits,
                                                                           int splits_b, int accumulate, int LS,
                                                                           const WgFoldJobs J)
{
    gw = J.njobs > 1 ? J.gw[blockIdx.y] : gw;
    part += (size_t)blockIdx.y * splits * ((size_t)Co * R);
    part_b += (size_t)blockIdx.y * splits_b * Co;
    wgrad_reduce_bn_body(part, gw, R, Co, part_b, splits, splits_b, accumulate, LS, J.f[blockIdx.y], blockIdx.x);
}

// ---- deferred reduces (conv.hip: lsn_wgrad_defer / lsn_wgrad_flush).  The 48 reduce launches of a training step read ~2 GB of
// partial tiles in kernels of 15 - 19 us each -- launch- and latency-bound at 1.8 TB/s, 1.2 ms per step with their gaps
// (profiles/r5_wgrad_reduce_cost.txt).  While a gradient arena collects the step's gradients (parallel/reducer.py) the calls leave
// their partial tiles where they are and queue a descriptor; one launch per kind reduces every queued gradient: the same
// arithmetic per element in the same order (the same bits), one ramp and one tail.
constexpr int RJ_MAX = 16;
struct RJobPlain {
    const float *part, *part_b;
    float *gw, *gb;
    unsigned long long n;
    int nb, splits, splits_b, LS, blk0, nblk;
};
struct RJobsPlain {
    RJobPlain j[RJ_MAX];
    int njobs;
};
static __global__ void conv_wgrad_reduce_multi_kernel(const RJobsPlain J)
{
    int ji = 0;
    while (ji + 1 < J.njobs && (int)blockIdx.x >= J.j[ji + 1].blk0) ++ji;
    const RJobPlain &q = J.j[ji];
    wgrad_reduce_body(q.part, q.gw, (size_t)q.n, q.part_b, q.gb, q.nb, q.splits, q.splits_b, 1, q.LS, blockIdx.x - q.blk0, q.nblk);
}
struct RJobFold {
    const float *part, *part_b;
    float *gw;
    WgFold f;
    int R, Co, splits, splits_b, LS, blk0;
};
struct RJobsFold {
    RJobFold j[RJ_MAX];
    int njobs;
};
static __global__ __launch_bounds__(256) void conv_wgrad_reduce_bn_multi_kernel(const RJobsFold J)
{
    int ji = 0;
    while (ji + 1 < J.njobs && (int)blockIdx.x >= J.j[ji + 1].blk0) ++ji;
    const RJobFold &q = J.j[ji];
    wgrad_reduce_bn_body(q.part, q.gw, q.R, q.Co, q.part_b, q.splits, q.splits_b, 1, q.LS, q.f, (int)blockIdx.x - q.blk0);
}

}  // namespace lsn
