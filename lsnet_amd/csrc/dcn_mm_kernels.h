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
// (dcn_gather_kernels.h: dcn_gather_kernel, dcn_anchor_sum_kernel, dcn_offgrad_kernel).
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

// Work distribution of a forward launch (dcn.hip dcn_sk_plan), the scheme of conv_kernels.h: workgroups [0, n_dp) take one whole
// (pixel tile, column block) each; workgroups [n_dp, n_dp + sk_n) share the LAST sk_tiles tiles evenly by chunks.  A tower
// launch of the LSNet step has 700 tiles for 512 resident workgroups (two rounds, the second 37 % full), a pyramid launch 2 100
// (five rounds, the last 10 % full).  A piece that holds part of a tile's sum leaves its accumulators in a slot, draws a ticket,
// and the piece that draws the last ticket adds the slots in chunk order (fixed order: bit-reproducible) and runs the epilogue.
struct DcnSk {
    int n_dp, sk_n, sk_tiles;
    float *part;
    unsigned *cnt;   // one counter per stream-K tile, zero between launches
};

// The hand-over of conv_mm_kernel's stream-K pieces as a function (agent-scope slot accesses, no fence: conv_kernels.h).
// Returns true in the workgroup that finishes tile kt (acc then holds the whole sum).
template <int TM, int TN>
__device__ __forceinline__ bool sk_hand_over(f32x16 (&acc)[TN][TM], const DcnSk &sk, int kt, int sk_s, bool opens_piece, int sk_U,
                                             int Tall, int tid, unsigned *ticket_lds)
{
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int SLOT = TM * TN * 1024 * 4;   // floats of a workgroup tile; element (q, tid) = float4 number q of thread tid
    constexpr int SC1 = 16;                    // agent scope (cache-policy bit 4 of the raw buffer intrinsics)
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(sk.part, 0, 2 * sk.sk_n * SLOT * 4, 0x00020000);
    {
        const int soff = (2 * sk_s + (opens_piece ? 0 : 1)) * (SLOT * 4);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), prs, (((j * TM + i) * 4 + g) * 256 + tid) * 16,
                                                           soff, SC1);
                }
    }
    const int u_lo = kt * Tall, u_hi = u_lo + Tall - 1;
    const int s_lo = (int)(((unsigned)(u_lo + 1) * (unsigned)sk.sk_n + (unsigned)sk_U - 1u) / (unsigned)sk_U) - 1;
    const int s_hi = (int)(((unsigned)(u_hi + 1) * (unsigned)sk.sk_n + (unsigned)sk_U - 1u) / (unsigned)sk_U) - 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's slot elements have reached the coherence point
    __syncthreads();
    if (tid == 0) *ticket_lds = __hip_atomic_fetch_add(sk.cnt + kt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if ((int)*ticket_lds != s_hi - s_lo) return false;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
    for (int c = s_lo; c <= s_hi; ++c) {   // chunk order = piece order
        const int cs = (int)((unsigned)sk_U * (unsigned)c / (unsigned)sk.sk_n);
        const int soff = (2 * c + (cs >= u_lo ? 0 : 1)) * (SLOT * 4);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f32x4 pv[TM * 4];
#pragma unroll
            for (int q = 0; q < TM * 4; ++q)
                pv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, ((j * TM * 4 + q) * 256 + tid) * 16, soff, SC1));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j][i][4 * g + e] += pv[i * 4 + g][e];
        }
    }
    if (tid == 0) __hip_atomic_store(sk.cnt + kt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed
    return true;
}

// FINE (experiment, LSNET_DCN_FWD_FINE=1): as in conv_mm_kernel -- the slice commit without its branch + sched_group_barrier
// groups put the blend / split instructions of a slice between its MFMAs instead of behind them.
// SK: the launch has stream-K pieces (DcnSk); a template parameter as in conv_mm_kernel.
template <int TM, int TN, int WM, int WN, int NP, bool FINE = false, bool SK = false>
__global__ __launch_bounds__(256, 2) void dcn_fwd_mm_kernel(const DcnArgs a, const unsigned short *__restrict__ wf,
                                                            int wf_bytes, const DcnSk sk)
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
    const int ncc = a.C / 32, NT = cv_nt(a.Co);
    const int Tall = K * ncc;
    const int ccpd = ncc / a.dg;   // chunks per deformable group
    // ---- this workgroup's share: one whole tile, or a stream-K piece ----
    const bool is_sk = SK && (int)blockIdx.x >= sk.n_dp;
    int sk_u = 0, sk_end = 0, sk_start = 0, sk_U = 0, sk_s = 0;
    if (is_sk) {
        sk_s = xcd_remap((int)blockIdx.x - sk.n_dp, sk.sk_n);
        sk_U = sk.sk_tiles * Tall;
        sk_start = sk_u = (int)((unsigned)sk_U * (unsigned)sk_s / (unsigned)sk.sk_n);
        sk_end = (int)((unsigned)sk_U * (unsigned)(sk_s + 1) / (unsigned)sk.sk_n);
    }
    __shared__ unsigned sk_ticket;
  for (;;) {   // one pass per segment (whole tiles: exactly one)
    int work, t_begin = 0, T = Tall;
    if (!is_sk) {
        work = xcd_remap((int)blockIdx.x, SK ? sk.n_dp : a.ntiles * ncb);
    } else {
        const int kt = (int)((unsigned)sk_u / (unsigned)Tall);
        work = sk.n_dp + kt;
        t_begin = sk_u - kt * Tall;
        const int te = sk_end - kt * Tall;
        T = (te < Tall ? te : Tall) - t_begin;
    }
    const bool sk_partial = SK && T != Tall;
    const int ptile = work / ncb;
    const Lvl &L = find_level(a, ptile);
    const int tile_p = (ptile - L.tile0) * BM;
    const int co_blk = (work - ptile * ncb) * BN;

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
    const int wsbase = (t_begin * NT + co_blk / 32) * (2 * NPL * 1024), wsstep = NT * (2 * NPL * 1024);
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
    if (SK) ci.k = t_begin / ncc, ci.cc = t_begin - ci.k * ncc;
    int kd_o = kd_of(ci), kd_w = kd_o;
    load_offsets(kd_o);
    load_wgts(kd_w);
    sx_soff = ci.cc * 128;
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

    // (round 5, as in conv_mm_kernel: the barrier sits in the middle of the iteration, the k-step-0 fragments of chunk t + 1 are
    // read right behind it under the MFMAs of k-step 1, the k-step-1 fragments of chunk t at the top under those of k-step 0)
    read_x(smem, 0);
    for (int t = 0; t < T; ++t) {
        const unsigned char *bc = smem + (t & 1) * BUF;
        unsigned char *bn = smem + ((t & 1) ^ 1) * BUF;
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
            if (FINE || t + 1 < T) commit_slice(ps, bn);   // (FINE: the last iteration commits a repeated chunk nobody reads)
            issue_slice(ps);
            if constexpr (FINE) {
#pragma unroll
                for (int g = 0; g < NM / NLD; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);   // vector / scalar ALU instructions
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        cm = ci;
        issue_w(t + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        read_x(bn, 0);   // (past the last chunk: never used)
        mfma_block(1);
        issue_w(t + 1, 1);
        __builtin_amdgcn_sched_barrier(0);
    }

    if (SK) __syncthreads();   // (the next segment rebuilds the table and buffer 0: every wave is past its last LDS read)
    bool do_out = true;
    if (SK && sk_partial) do_out = sk_hand_over<TM, TN>(acc, sk, work - sk.n_dp, sk_s, sk_u == sk_start, sk_U, Tall, tid, &sk_ticket);

    // ---- epilogue: lane = pixel (lane & 31) of tile i, output channels 8 g + 4 (lane >> 5) + (0..3) of tile j ----
    if (do_out) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int pix = tile_p + wm * TM * 32 + i * 32 + (lane & 31);
        if (pix >= L.P) continue;
        float *orow = L.out + (size_t)pix * a.opitch;
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
    if (!SK || !is_sk) break;
    sk_u += T;
    if (sk_u >= sk_end) break;
  }
}

// =============================================================================================
// Weight gradient   gw[co][k][c] = sum_p gout[p][co] * col[p][k, c]   (deform_conv_cuda.cpp:1126-1131): the reduction
// index is the pixel.
//   * grad_output plays the part the weights play in the forward kernels: dcn_gout_frag_kernel writes it ONCE per launch
//     as bf16 planes in MFMA fragment order ([16-pixel k-step][32-co tile][plane][lane: co = lane & 31, pixels
//     8 (lane >> 5) .. + 7]), and every workgroup of the 36 column blocks fetches its fragments from L2 straight into
//     registers -- instead of each of them loading, splitting and transposing the same fp32 rows through LDS.  The same
//     pass sums the bias gradient.
//   * the sampled columns (one tap, 64 channels per workgroup) are blended from their four corners, split and stored as
//     natural [pixel][channel] rows of 128 B; ds_read_b64_tr_b16 delivers the pixel-major fragments.
//   * workgroup = 256 co x 64 columns, wave = 64 x 64; 32-pixel chunks, two LDS stages, staging of chunk t + 1 and the
//     corner loads of chunk t + 2 between the MFMAs of chunk t (the schedule of conv_mm_kernel).
//   * the pixel range is split over blockIdx.y; each split stores its partial tile, conv_wgrad_reduce_kernel adds them in
//     a fixed order: no atomics, deterministic.
// Conditions (dcn.hip mm_wgrad_ok): groups == 1, 256 | Co, 64 | C / deformable_groups, the launch-wide sampling table of
// the backward-data pass (a.gtap).
// =============================================================================================
template <int NPL>
__global__ __launch_bounds__(256) void dcn_gout_frag_kernel(const DcnArgs a, int nsteps16, int steps_per_block,
                                                            unsigned short *__restrict__ img, float *__restrict__ part_b)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int NT = a.Co / 32;
    const int tile = blockIdx.y * 4 + wave;   // (Co % 128 == 0)
    const int co = tile * 32 + (lane & 31), h = lane >> 5;
    const int s0 = blockIdx.x * steps_per_block, s1 = min(s0 + steps_per_block, nsteps16);
    float bsum = 0.f;
    for (int s = s0; s < s1; ++s) {
        const Lvl &L = find_level(a, s >> 1);            // level tables count 32-pixel chunks
        const int p0 = (s - 2 * L.tile0) * 16 + 8 * h;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int pp = p0 + e;
            v[e] = pp < L.P ? L.gout[(size_t)pp * a.opitch + co] : 0.f;
            bsum += v[e];
        }
        unsigned pl[4][NPL];
#pragma unroll
        for (int e = 0; e < 4; ++e) split_planes<NPL>(v[2 * e], v[2 * e + 1], pl[e]);
        unsigned short *dst = img + ((size_t)s * NT + tile) * NPL * 512 + (size_t)lane * 8;
#pragma unroll
        for (int q = 0; q < NPL; ++q)
            *reinterpret_cast<uint4 *>(dst + (size_t)q * 512) = make_uint4(pl[0][q], pl[1][q], pl[2][q], pl[3][q]);
    }
    if (part_b != nullptr) {
        bsum += __shfl_xor(bsum, 32);
        if (lane < 32) part_b[(size_t)blockIdx.x * a.Co + co] = bsum;
    }
}

// The launch-wide sampling table (layout of dcn_bin_kernel: k-major, [kd * gtap_rows + prow0 + pix], the all-zero "no
// sample" entry behind it) for a launch without a backward-data pass: the weight gradient of a dense convolution (Lvl.off
// == NULL: the regular grid, every position integral) through dcn_wgrad_mm_kernel<NP, DENSE = true>.  The same launch
// writes the per-chunk table of dcn_chunk_meta_kernel (meta != NULL).
__device__ __forceinline__ void dcn_chunk_meta_item(const DcnArgs &a, int t, int *__restrict__ meta);
__global__ void dcn_tap_table_kernel(const DcnArgs a, int nentries, Tap *__restrict__ gtap, int nchunks, int *__restrict__ meta)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int K = a.kh * a.kw;
    if (e == 0) {
        Tap z = {};
        gtap[nentries] = z;
    }
    if (meta != nullptr && e < nchunks) dcn_chunk_meta_item(a, e, meta);
    if (e >= nentries) return;
    const int kd = e / a.gtap_rows, prow = e - kd * a.gtap_rows;
    const int dgi = kd / K, k = kd - dgi * K;
    int li = 0;
    while (li + 1 < a.nlv && prow >= a.lv[li + 1].prow0) ++li;
    const Lvl &L = a.lv[li];
    gtap[e] = make_tap(a, L, prow - L.prow0, k, dgi);
}

// Per 32-pixel chunk: where it lives -- {input base lo, hi, input bytes, first row of the launch-wide sampling table, valid
// pixels, -, -, -}.  The weight-gradient kernel reads it with scalar loads: a level lookup per iteration in the kernel
// itself (dynamic indexing of the by-value argument struct) compiled to vector loads from the argument copy with full
// vmcnt(0) waits in the middle of the software pipeline.
__device__ __forceinline__ void dcn_chunk_meta_item(const DcnArgs &a, int t, int *__restrict__ meta)
{
    const Lvl &L = find_level(a, t);
    const unsigned long long p = reinterpret_cast<unsigned long long>(L.x);
    const int p0 = (t - L.tile0) * 32;
    int4 m0, m1;
    m0.x = (int)(unsigned)p, m0.y = (int)(unsigned)(p >> 32), m0.z = L.B * L.H * L.W * a.C * 4, m0.w = L.prow0 + p0;
    m1.x = max(0, min(32, L.P - p0)), m1.y = m1.z = m1.w = 0;
    reinterpret_cast<int4 *>(meta)[2 * t] = m0;
    reinterpret_cast<int4 *>(meta)[2 * t + 1] = m1;
}
__global__ void dcn_chunk_meta_kernel(const DcnArgs a, int nchunks, int *__restrict__ meta)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nchunks) dcn_chunk_meta_item(a, t, meta);
}

__host__ __device__ inline size_t dcn_wgrad_mm_lds_bytes(int npl) { return (size_t)2 * npl * 32 * 128 + 4 * 32 * sizeof(Tap); }

// DENSE: every sampling position is a grid point (weights 1, 0, 0, 0): one load per position instead of four, and the value
// itself (times the validity of the position: zero padding) instead of the blend.
// FINE (experiment, LSNET_DCN_WGRAD_FINE=1): the slice commit without its `t + 1 < T` branch (the last iteration commits a
// repeated chunk into the stage nobody reads again) + sched_group_barrier groups, so that the ~75 blend / split
// instructions of a slice sit BETWEEN its 12 MFMAs instead of behind them (see conv_mm_kernel).
template <int NP, bool DENSE = false, bool FINE = false>
__global__ __launch_bounds__(256, 2) void dcn_wgrad_mm_kernel(const DcnArgs a, int nchunks, const unsigned short *__restrict__ gimg,
                                                              int gimg_bytes, float *__restrict__ part, const int *__restrict__ meta)
{
    using SC = SplitCfg<NP>;
    constexpr int NPL = SC::NPL;
    constexpr int TI = 2, TJ = 2;                  // wave: 64 columns x 64 output channels
    constexpr int RBX = 128, XPL = 32 * RBX, STAGE = NPL * XPL;
    extern __shared__ __align__(16) unsigned char smem[];   // 2 x STAGE, then the sampling table: 4 slots x 32 entries
    Tap *tab = reinterpret_cast<Tap *>(smem + 2 * STAGE);

    const int tid = threadIdx.x, lane = tid & 63, wj = tid >> 6;
    const int K = a.kh * a.kw, KD = K * a.dg;
    const int ncolb = a.C / 64;                    // column blocks per tap
    const int ncol = (int)gridDim.x, nsplit = (int)gridDim.y;
    const int work = xcd_remap(blockIdx.y * ncol + blockIdx.x, ncol * nsplit);
    const int split = work / ncol, bcol = work - split * ncol;
    const int k = bcol / ncolb, c0 = (bcol - k * ncolb) * 64;
    const int kd = (c0 / (a.C / a.dg)) * K + k;
    const int co_blk = blockIdx.z * 256;
    const int NT = a.Co / 32;
    // (readfirstlane: the 64-bit divisions run on the vector ALU; left in VGPRs, every level lookup below became a
    // vector load from the kernel arguments with a full vmcnt(0) wait in the middle of the software pipeline)
    // (32-bit quotients: the host checks nchunks * nsplit < 2^32)
    const int t_begin = __builtin_amdgcn_readfirstlane((int)((unsigned)nchunks * (unsigned)split / (unsigned)nsplit));
    const int T = __builtin_amdgcn_readfirstlane((int)((unsigned)nchunks * (unsigned)(split + 1) / (unsigned)nsplit)) - t_begin;

    const __amdgpu_buffer_rsrc_t grs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(gimg), 0, gimg_bytes, 0x00020000);

    // ---- sampling table: the chunk's 32 entries are 1 KB contiguous in the k-major launch-wide table; lane l of wave 0
    // moves 16-byte half (l & 1) of entry (l >> 1); pixels past the level's end take the all-zero entry behind the table
    typedef const int __attribute__((address_space(4))) *cmeta_t;   // constant address space: uniform address -> s_load
    cmeta_t cmeta = (cmeta_t)(unsigned long long)meta;
    auto gtap_load = [&](int t) -> uint4 {
        const int grow = cmeta[8 * t + 3], valid = cmeta[8 * t + 4];
        const int pix = lane >> 1;
        const size_t ent = pix < valid ? (size_t)kd * a.gtap_rows + grow + pix : (size_t)KD * a.gtap_rows;
        return reinterpret_cast<const uint4 *>(a.gtap)[ent * 2 + (lane & 1)];
    };
    auto gtap_put = [&](int slot, uint4 v) { reinterpret_cast<uint4 *>(tab + slot * 32)[lane] = v; };

    // ---- sampled columns: thread = (float4 slot xc4 of the 64-channel slab, pixel q of a 16-pixel pass), two passes ----
    const int xc4 = tid & 15, xq = tid >> 4;
    int xlds[2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int pix = ps * 16 + xq;
        const int sl = (xc4 >> 2) ^ (2 * ((pix >> 1) & 1));   // 32-byte slot swizzle of the 128-byte rows
        xlds[ps] = pix * RBX + sl * 32 + (xc4 & 3) * 8;
    }
    float4 xv[2][4];
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.lv[0].x), 0, 0, 0x00020000);
    auto open_chunk = [&](int t) {   // buffer descriptor of the level chunk t lies in
        // (readfirstlane: hipcc kept a descriptor that changes inside the loop in VGPRs and wrapped every corner load in a
        // waterfall loop)
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)cmeta[8 * t]), hi = __builtin_amdgcn_readfirstlane((unsigned)cmeta[8 * t + 1]);
        const int nrec = __builtin_amdgcn_readfirstlane(cmeta[8 * t + 2]);
        xrs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float *>(((unsigned long long)hi << 32) | lo), 0, nrec, 0x00020000);
    };
    // Round 6: inside the loop the descriptor is fetched in two halves.  Its three scalars are REQUESTED behind the barrier in
    // the middle of iteration t - 1 and turned into the descriptor at the top of iteration t, before the fragment reads: a
    // scalar load can only be waited for with lgkmcnt(0), and requested at the top it drained the twelve fragment reads
    // issued just before it in front of the first MFMA of every chunk.
    unsigned pd_lo = 0, pd_hi = 0;
    int pd_nrec = 0;
    auto request_chunk = [&](int t) { pd_lo = (unsigned)cmeta[8 * t], pd_hi = (unsigned)cmeta[8 * t + 1], pd_nrec = cmeta[8 * t + 2]; };
    auto take_chunk = [&]() {
        const unsigned lo = __builtin_amdgcn_readfirstlane(pd_lo), hi = __builtin_amdgcn_readfirstlane(pd_hi);
        const int nrec = __builtin_amdgcn_readfirstlane(pd_nrec);
        xrs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float *>(((unsigned long long)hi << 32) | lo), 0, nrec, 0x00020000);
    };
    auto issue_slice = [&](int ps, int slot) {
        const int4 idx = *reinterpret_cast<const int4 *>(&tab[slot * 32 + ps * 16 + xq]);
        const int cq = (c0 + 4 * xc4) * 4;
        xv[ps][0] = cv_load4(xrs, idx.x * 4 + cq, 0);
        if (!DENSE) {
            xv[ps][1] = cv_load4(xrs, idx.y * 4 + cq, 0);
            xv[ps][2] = cv_load4(xrs, idx.z * 4 + cq, 0);
            xv[ps][3] = cv_load4(xrs, idx.w * 4 + cq, 0);
        }
    };
    auto commit_slice = [&](int ps, int slot, unsigned char *buf) {
        const Tap tp = tab[slot * 32 + ps * 16 + xq];
        float v[4];
        if (DENSE) {
            const bool in = (tp.flags & 1) != 0;   // a padding position (or a pixel past the level's end): zero
            v[0] = in ? xv[ps][0].x : 0.f, v[1] = in ? xv[ps][0].y : 0.f;
            v[2] = in ? xv[ps][0].z : 0.f, v[3] = in ? xv[ps][0].w : 0.f;
        } else {
            float b00, b01, b10, b11;
            corner_weights(tp, b00, b01, b10, b11);
            b00 *= tp.m, b01 *= tp.m, b10 *= tp.m, b11 *= tp.m;
            v[0] = b00 * xv[ps][0].x + b01 * xv[ps][1].x + b10 * xv[ps][2].x + b11 * xv[ps][3].x;
            v[1] = b00 * xv[ps][0].y + b01 * xv[ps][1].y + b10 * xv[ps][2].y + b11 * xv[ps][3].y;
            v[2] = b00 * xv[ps][0].z + b01 * xv[ps][1].z + b10 * xv[ps][2].z + b11 * xv[ps][3].z;
            v[3] = b00 * xv[ps][0].w + b01 * xv[ps][1].w + b10 * xv[ps][2].w + b11 * xv[ps][3].w;
        }
        unsigned p0[NPL], p1[NPL];
        split_planes<NPL>(v[0], v[1], p0);
        split_planes<NPL>(v[2], v[3], p1);
#pragma unroll
        for (int q = 0; q < NPL; ++q) *reinterpret_cast<uint2 *>(buf + q * XPL + xlds[ps]) = make_uint2(p0[q], p1[q]);
    };

    // ---- grad_output fragments: chunk t, k-step ks, tile, plane: (((t * 2 + ks) * NT + tile) * NPL + q) * 1024 + lane * 16
    const int gvoff = lane * 16 + (co_blk / 32 + wj * TJ) * NPL * 1024;
    const int gsstep = NT * NPL * 1024;
    bf16x8 Gf[2][TJ][NPL];
    auto issue_g = [&](int t, int ks) {   // t saturates at the last chunk (a repeated L2 hit, never used)
        const int soff = __builtin_amdgcn_readfirstlane(((t_begin + (t < T ? t : T - 1)) * 2 + ks) * gsstep);
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int q = 0; q < NPL; ++q) Gf[ks][j][q] = cv_load_frag(grs, gvoff + (j * NPL + q) * 1024, soff);
    };

    // ---- column fragment addresses: lane: column group g of the 32-wide tile, 8-byte piece cq of block row rb; pixels
    // m = 8 (lane >> 5) + 4 h + rb of k-step ks for the two tr-reads h = 0, 1 ----
    const int fg = (lane >> 4) & 1, frb = (lane >> 2) & 3, fcq = lane & 3;
    int xaddr[2][TI][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int pp = ks * 16 + 8 * (lane >> 5) + 4 * h + frb;
                const int sl = (2 * i + fg) ^ (2 * ((pp >> 1) & 1));
                xaddr[ks][i][h] = pp * RBX + sl * 32 + fcq * 8;
            }
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

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (T > 0) {
        // ---- prologue: table slots of chunks 0 .. 2 (slot = chunk % 4; chunk 3 is in flight); chunk 0 -> stage 0; corners
        // of chunk 1 and the grad_output fragments of chunk 0 in flight ----
        auto sat = [&](int t) { return __builtin_amdgcn_readfirstlane(t_begin + (t < T ? t : T - 1)); };
        if (wj == 0) {
            gtap_put(0, gtap_load(sat(0)));
            gtap_put(1, gtap_load(sat(1)));
            gtap_put(2, gtap_load(sat(2)));
        }
        uint4 tq = gtap_load(sat(3));
        __syncthreads();
        open_chunk(sat(0));
        issue_slice(0, 0);
        issue_slice(1, 0);
        commit_slice(0, 0, smem);
        commit_slice(1, 0, smem);
        open_chunk(sat(1));
        issue_slice(0, 1);
        issue_slice(1, 1);
        issue_g(0, 0);
        issue_g(0, 1);
        __syncthreads();

        // (round 5, as in conv_mm_kernel: the barrier sits in the MIDDLE of the iteration; the k-step-0 fragments of chunk t + 1
        // are read right behind it under the MFMAs of k-step 1, the k-step-1 fragments of chunk t at the top)
        bf16x8 Xf[2][TI][NPL];
        auto read_x = [&](const unsigned char *buf, int ks) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int q = 0; q < NPL; ++q)
                    Xf[ks][i][q] = frag(buf + q * XPL + xaddr[ks][i][0], buf + q * XPL + xaddr[ks][i][1]);
        };
        read_x(smem, 0);
        request_chunk(sat(2));
        for (int t = 0; t < T; ++t) {
            const unsigned char *bc = smem + (t & 1) * STAGE;
            unsigned char *bn = smem + ((t & 1) ^ 1) * STAGE;
            const int slot1 = (t + 1) & 3, slot2 = (t + 2) & 3, slot3 = (t + 3) & 3;
            take_chunk();   // chunk t + 2 (requested half an iteration ago)
            __builtin_amdgcn_sched_barrier(0);
            // slot3 held chunk t - 1: last read in iteration t - 2 (commit) -- two barriers ago; its new entries (chunk t + 3)
            // are first read in iteration t + 1 (issue_slice), behind the barrier in the middle of this one
            if (wj == 0) gtap_put(slot3, tq);
            tq = gtap_load(sat(t + 4));
            read_x(bc, 1);
            __builtin_amdgcn_sched_barrier(0);
            constexpr int NM = NP * TI * TJ;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
#pragma unroll
                for (int m = ps * NM / 2; m < (ps + 1) * NM / 2; ++m) {
                    const int prod = m / (TI * TJ), i = (m / TJ) % TI, j = m % TJ;
                    acc[i][j] = mfma_bf16(Xf[0][i][SC::pa(prod)], Gf[0][j][SC::pb(prod)], acc[i][j]);
                }
                if (FINE || t + 1 < T) commit_slice(ps, slot1, bn);   // registers hold the corners of chunk t + 1
                issue_slice(ps, slot2);                       // chunk t + 2 (saturated: a repeated fetch, never committed)
                if constexpr (FINE) {
#pragma unroll
                    for (int g = 0; g < NM / 2; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x006, DENSE ? 4 : 6, 0);      // vector / scalar ALU instructions
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            issue_g(t + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            request_chunk(sat(t + 3));
            __builtin_amdgcn_sched_barrier(0);
            read_x(bn, 0);   // (past the last chunk: never used)
#pragma unroll
            for (int prod = 0; prod < NP; ++prod)
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = mfma_bf16(Xf[1][i][SC::pa(prod)], Gf[1][j][SC::pb(prod)], acc[i][j]);
            issue_g(t + 1, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue: D[column][co]: lane = co (lane & 31) of tile j, columns 8 q + 4 (lane >> 5) + (0..3) of tile i ----
    const size_t nW = (size_t)a.Co * K * a.C;
    float *pw = part + (size_t)split * nW;
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int co = co_blk + (wj * TJ + j) * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = c0 + i * 32 + 8 * q + 4 * (lane >> 5);
                *reinterpret_cast<float4 *>(pw + ((size_t)co * K + k) * a.C + c) =
                    make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
            }
    }
}

}  // namespace lsn
