// Deformable convolution on the dense-convolution skeleton of conv_kernels.h (two workgroups per CU, weights as MFMA
// fragments straight from L2, pixel operand split into bf16 planes in a double-buffered 64-byte-row LDS image, D rows =
// channels so that the epilogue moves 16 bytes per lane).  Included by dcn.hip.
//
//   forward        out[p][co]      = sum_{k, c} col[p][k, c] * w[co][k, c],   col = mask * bilinear(x, sample(p, k))
//                  (deform_conv_cuda_kernel.cu:246-297 + the addmm_ of deform_conv_cuda.cpp:662-684, fused)
//
// It is the dense kernel with ONE phase replaced: the pixel operand of chunk (tap k, 32 channels) is not one float4 per
// lane but the bilinear blend of four (the sample's corners; corner offsets and mask-weighted corner weights come from a
// per-workgroup table in LDS that is built once per 64-pixel tile).
// Conditions (dcn.hip mm_fwd_ok): groups == 1, 32 | C / deformable_groups, 128 | Co; everything else stays on the
// kernels of dcn_kernels.h.
//
// The backward-data GEMM  gcol[p][k, c] = sum_co gout[p][co] * w[co][k, c]  (deform_conv_cuda.cpp:783-787) needs no
// kernel of its own: it IS conv_mm_kernel as a 1x1 convolution of grad_output with N = K * C output columns
// (conv.hip conv_mm_rows), writing the column gradients unweighted; the modulation scalar, the scatter into grad_input
// and the corner sums of grad_offset / grad_mask all belong to the gather pass that reads those rows anyway
// (dcn_kernels.h: dcn_gather_kernel, dcn_anchor_sum_kernel, dcn_offgrad_kernel).
#pragma once
#include "conv_kernels.h"
#include "dcn_kernels.h"

namespace lsn {

// forward sampling-table entry: clamped corner offsets (floats, channel 0) and mask-weighted corner weights
struct __align__(16) FTap {
    int i[4];
    float w[4];
};

__host__ __device__ inline size_t dcn_fwd_mm_lds_bytes(int npl, int KD) { return (size_t)2 * npl * 64 * 64 + (size_t)64 * KD * sizeof(FTap); }

template <int TM, int TN, int WM, int WN, int NP>
__global__ __launch_bounds__(256, 2) void dcn_fwd_mm_kernel(const DcnArgs a, const unsigned short *__restrict__ wf,
                                                            int wf_bytes)
{
    using SC = SplitCfg<NP>;
    constexpr int NPL = SC::NPL;
    static_assert(WM * WN == 4 && WM * TM == 2, "four waves, 64-pixel tiles");
    constexpr int BM = 64, BN = WN * TN * 32;
    constexpr int NLD = BM / 32;
    constexpr int PLANE = BM * 64, BUF = NPL * PLANE;
    extern __shared__ __align__(16) unsigned char smem[];   // 2 x BUF, then the sampling table
    FTap *tab = reinterpret_cast<FTap *>(smem + 2 * BUF);    // [64 px][K * dg]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int K = a.kh * a.kw, KD = K * a.dg;
    const int ncb = a.Co / BN;
    const int work = xcd_remap(blockIdx.x, a.ntiles * ncb);
    const int ptile = work / ncb;
    const Lvl &L = find_level(a, ptile);
    const int tile_p = (ptile - L.tile0) * BM;
    const int co_blk = (work - ptile * ncb) * BN;
    const int ncc = a.C / 32, NT = cv_nt(a.Co);
    const int T = K * ncc;
    const int ccpd = ncc / a.dg;   // chunks per deformable group

    for (int e = tid; e < BM * KD; e += 256) {
        const int pl = e / KD, r = e - pl * KD;
        const int dgi = r / K, k = r - dgi * K;
        const Tap tp = make_tap(a, L, tile_p + pl, k, dgi);
        float b00, b01, b10, b11;
        corner_weights(tp, b00, b01, b10, b11);
        FTap f;
        f.i[0] = tp.i00, f.i[1] = tp.i01, f.i[2] = tp.i10, f.i[3] = tp.i11;
        f.w[0] = b00 * tp.m, f.w[1] = b01 * tp.m, f.w[2] = b10 * tp.m, f.w[3] = b11 * tp.m;
        tab[e] = f;
    }

    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(L.x), 0, L.B * L.H * L.W * a.C * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(wf), 0, wf_bytes, 0x00020000);

    // ---- pixel operand: thread = (float4 slot c4 of the 32-channel slab, pixel row prow of a 32-row pass) ----
    const int c4 = tid & 7, prow = tid >> 3;
    const int st_off = prow * 64 + ((((c4 >> 1) ^ ((prow >> 2) & 3)) << 4) | ((c4 & 1) << 3));
    struct Ck {
        int k, cc;
    };
    auto next = [&](Ck &c) {
        if (++c.cc == ncc) {
            c.cc = 0;
            ++c.k;
        }
    };
    auto kd_of = [&](const Ck &c) { return (a.dg == 1 ? 0 : (c.cc / ccpd) * K) + c.k; };
    int voff[NLD][4];     // corner byte offsets of the chunk under issue (+ this thread's channel slot)
    float wgt[NLD][4];    // corner weights of the chunk being committed
    float4 xv[NLD][4];
    auto load_offsets = [&](int kd) {
#pragma unroll
        for (int ps = 0; ps < NLD; ++ps) {
            const int4 idx = *reinterpret_cast<const int4 *>(&tab[(ps * 32 + prow) * KD + kd].i[0]);
            voff[ps][0] = (idx.x + 4 * c4) * 4, voff[ps][1] = (idx.y + 4 * c4) * 4;
            voff[ps][2] = (idx.z + 4 * c4) * 4, voff[ps][3] = (idx.w + 4 * c4) * 4;
        }
    };
    auto load_wgts = [&](int kd) {
#pragma unroll
        for (int ps = 0; ps < NLD; ++ps) {
            const float4 w = *reinterpret_cast<const float4 *>(&tab[(ps * 32 + prow) * KD + kd].w[0]);
            wgt[ps][0] = w.x, wgt[ps][1] = w.y, wgt[ps][2] = w.z, wgt[ps][3] = w.w;
        }
    };
    int sx_soff = 0;   // scalar byte offset of the chunk under issue: its 32-channel slab
    auto issue_slice = [&](int ps) {
#pragma unroll
        for (int q = 0; q < 4; ++q) xv[ps][q] = cv_load4(xrs, voff[ps][q], sx_soff);
    };
    auto commit_slice = [&](int ps, unsigned char *buf) {
        float v[4];
        v[0] = wgt[ps][0] * xv[ps][0].x + wgt[ps][1] * xv[ps][1].x + wgt[ps][2] * xv[ps][2].x + wgt[ps][3] * xv[ps][3].x;
        v[1] = wgt[ps][0] * xv[ps][0].y + wgt[ps][1] * xv[ps][1].y + wgt[ps][2] * xv[ps][2].y + wgt[ps][3] * xv[ps][3].y;
        v[2] = wgt[ps][0] * xv[ps][0].z + wgt[ps][1] * xv[ps][1].z + wgt[ps][2] * xv[ps][2].z + wgt[ps][3] * xv[ps][3].z;
        v[3] = wgt[ps][0] * xv[ps][0].w + wgt[ps][1] * xv[ps][1].w + wgt[ps][2] * xv[ps][2].w + wgt[ps][3] * xv[ps][3].w;
        unsigned p0[NPL], p1[NPL];
        split_planes<NPL>(v[0], v[1], p0);
        split_planes<NPL>(v[2], v[3], p1);
        unsigned char *p = buf + ps * 32 * 64 + st_off;
#pragma unroll
        for (int q = 0; q < NPL; ++q) *reinterpret_cast<uint2 *>(p + q * PLANE) = make_uint2(p0[q], p1[q]);
    };

    // ---- weight operand: fragments straight from L2 (conv_kernels.h) ----
    const int wvoff = lane * 16 + wn * TN * (2 * NPL * 1024);
    const int wsbase = (co_blk / 32) * (2 * NPL * 1024), wsstep = NT * (2 * NPL * 1024);
    bf16x8 Wf[2][TN][NPL];
    auto issue_w = [&](int t, int ks) {   // t saturates at the last chunk (a repeated L2 hit, never used)
        const int soff = wsbase + (t < T ? t : T - 1) * wsstep;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < NPL; ++q)
                Wf[ks][j][q] = cv_load_frag(wrs, wvoff + ((j * 2 + ks) * NPL + q) * 1024, soff);
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

    __syncthreads();   // sampling table complete

    // ---- prologue: chunk 0 -> LDS buffer 0, raw corners of chunk 1 and the weight fragments of chunk 0 in flight ----
    Ck ci = {0, 0};        // chunk under issue
    int kd_o = kd_of(ci), kd_w = kd_o;
    load_offsets(kd_o);
    load_wgts(kd_w);
#pragma unroll
    for (int ps = 0; ps < NLD; ++ps) issue_slice(ps);
#pragma unroll
    for (int ps = 0; ps < NLD; ++ps) commit_slice(ps, smem);
    if (T > 1) next(ci);
    if (kd_of(ci) != kd_o) {
        kd_o = kd_of(ci);
        load_offsets(kd_o);
    }
    sx_soff = ci.cc * 128;
#pragma unroll
    for (int ps = 0; ps < NLD; ++ps) issue_slice(ps);
    Ck cm = ci;            // chunk whose raw corners sit in xv (committed next)
    issue_w(0, 0);
    issue_w(0, 1);
    __syncthreads();

    const int frow = wm * TM * 32 + (lane & 31);
    const int fsw = (frow >> 2) & 3;
    bf16x8 Xf[2][TM][NPL];
    auto read_x = [&](const unsigned char *buf, int ks) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < NPL; ++q)
                Xf[ks][i][q] = *reinterpret_cast<const bf16x8 *>(buf + q * PLANE + (frow + i * 32) * 64 +
                                                                 (((ks * 2 + (lane >> 5)) ^ fsw) << 4));
    };
    auto mfma_block = [&](int ks) {
#pragma unroll
        for (int prod = 0; prod < NP; ++prod)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc[j][i] = mfma_bf16(Wf[ks][j][SC::pb(prod)], Xf[ks][i][SC::pa(prod)], acc[j][i]);
    };

    for (int t = 0; t < T; ++t) {
        const unsigned char *bc = smem + (t & 1) * BUF;
        unsigned char *bn = smem + ((t & 1) ^ 1) * BUF;
        read_x(bc, 0);
        read_x(bc, 1);
        {   // weights of the chunk in xv (t + 1), offsets of the chunk to fetch (t + 2)
            const int kdw = kd_of(cm);
            if (kdw != kd_w) {
                kd_w = kdw;
                load_wgts(kd_w);
            }
            if (t + 2 < T) next(ci);
            const int kdo = kd_of(ci);
            if (kdo != kd_o) {
                kd_o = kdo;
                load_offsets(kd_o);
            }
            sx_soff = ci.cc * 128;
        }
        __builtin_amdgcn_sched_barrier(0);
        constexpr int NM = NP * TN * TM;
#pragma unroll
        for (int ps = 0; ps < NLD; ++ps) {
#pragma unroll
            for (int m = ps * NM / NLD; m < (ps + 1) * NM / NLD; ++m) {
                const int prod = m / (TN * TM), j = (m / TM) % TN, i = m % TM;
                acc[j][i] = mfma_bf16(Wf[0][j][SC::pb(prod)], Xf[0][i][SC::pa(prod)], acc[j][i]);
            }
            if (t + 1 < T) commit_slice(ps, bn);
            issue_slice(ps);
            __builtin_amdgcn_sched_barrier(0);
        }
        cm = ci;
        issue_w(t + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_block(1);
        issue_w(t + 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }

    // ---- epilogue: lane = pixel (lane & 31) of tile i, output channels 8 g + 4 (lane >> 5) + (0..3) of tile j ----
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int pix = tile_p + wm * TM * 32 + i * 32 + (lane & 31);
        if (pix >= L.P) continue;
        float *orow = L.out + (size_t)pix * a.Co;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = co_blk + (wn * TN + j) * 32 + 8 * g + 4 * (lane >> 5);
                float4 v = make_float4(acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]);
                if (a.bias) {
                    const float4 bv = *reinterpret_cast<const float4 *>(a.bias + co);
                    v.x += bv.x, v.y += bv.y, v.z += bv.z, v.w += bv.w;
                }
                *reinterpret_cast<float4 *>(orow + co) = v;
            }
    }
}

}  // namespace lsn
