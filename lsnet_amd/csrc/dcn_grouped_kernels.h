// Grouped deformable convolution (ResNeXt-101 64x4d-DCN, BASELINE config 4: 64 groups of 8 / 16 / 32 channels in c3 - c5;
// resnext.py:11-83, deform_conv_cuda.cpp:613-692 runs the groups one after the other): forward and weight gradient.
// Included by dcn.hip.
//
// Round 6 (profiles/r6_cfg4_kernel_stats.txt): the config-4 step spent 71 of its 203 ms in the weight gradient of these layers
// (dcn_wgrad_xn_kernel: 256 x 64 MFMA tiles of which a group fills 1/16 x 1/4) and 30 ms in their forward (dcn_fwd_kernel:
// 64 x 64 fp32-MFMA tiles per group, 64-byte gathers).  Per group the product is 8 .. 32 wide, 2.5 GFLOP per layer in all: the
// work is the GATHER (1.2 GB of corner reads per layer through the L1s) and it belongs to whoever reads contiguous channels.
//
//   * the sampled columns col[pixel][tap][channel] of a pixel tile are blended ONCE per workgroup, by threads that walk the
//     channels (a corner of a tap is one run of 64 .. 256 contiguous bytes), into LDS;
//   * forward: per group out[px][co] = col[px][tap, ci] . w[co][tap, ci] on the fp32 matrix instructions
//     (v_mfma_f32_16x16x4_f32: exact fp32, the fmaf chain of the reference, in every math mode; operands come from LDS once per
//     fragment instead of once per product);
//   * weight gradient: a thread owns one (co, ci) pair of a group and its nine taps, walks the pixel range of its split over
//     the staged columns and grad_output rows (exact fp32 fmaf chains); splits leave partial tiles that the ordered reduce of
//     conv_wgrad_kernels.h adds up: deterministic, deferrable like every other weight gradient.
#pragma once
#include "dcn_kernels.h"

namespace lsn {

constexpr int GRP_PB = 16;   // pixels per staged chunk of the weight-gradient kernel

// channels a 256-pair block of the weight-gradient kernel covers: NCO output, NCH input
template <int CG> struct GrpW {
    static constexpr int PAIRS = CG * CG;
    static constexpr int NCO = 256 / CG;
    static constexpr int NCH = CG >= 16 ? CG : 256 / CG;
};

__host__ __device__ inline size_t dcn_wgrad_grouped_lds_bytes(int K, int nch, int nco)
{
    return ((size_t)GRP_PB * K * nch + (size_t)GRP_PB * nco) * sizeof(float);
}

// one sampling position of launch-wide pixel row `prow` (level L), tap kd: from the backward-data pass's table when there is one
__device__ __forceinline__ Tap grp_tap(const DcnArgs &a, const Lvl &L, int prow, int k, int dgi)
{
    if (a.gtap != nullptr) return a.gtap[(size_t)(dgi * a.kh * a.kw + k) * a.gtap_rows + prow];
    return make_tap(a, L, prow - L.prow0, k, dgi);
}

// gw[g CG + co][k][ci] = sum_p gout[p][g CG + co] * col[p][k][g CG + ci]   (deform_conv_cuda.cpp:1126-1131 per group)
// grid: (pixel splits, pair blocks); part[split][Co K CG], part_b[split][Co]
template <int CG, int KMAX>
__global__ __launch_bounds__(256) void dcn_wgrad_grouped_kernel(const DcnArgs a, int nrows, int rows_per_split, float *__restrict__ part,
                                                                float *__restrict__ part_b)
{
    using G = GrpW<CG>;
    constexpr int NCO = G::NCO, NCH = G::NCH, Q4 = NCH / 4;
    extern __shared__ __align__(16) float gsm[];
    const int K = a.kh * a.kw;
    float *col_s = gsm;                               // [GRP_PB][K][NCH]
    float *gout_s = gsm + GRP_PB * K * NCH;            // [GRP_PB][NCO]
    const int tid = threadIdx.x;
    const int q = blockIdx.y * 256 + tid;
    const int g = q / G::PAIRS, r = q - g * G::PAIRS, co = r / CG, ci = r - co * CG;
    const int q0 = blockIdx.y * 256, g0 = q0 / G::PAIRS;
    const int cin0 = g0 * CG;
    const int cout0 = g0 * CG + (q0 - g0 * G::PAIRS) / CG;
    const int lci = g * CG + ci - cin0, lco = g * CG + co - cout0;
    const int cpdg = a.C / a.dg, dgi = cin0 / cpdg;    // (the block's channels lie in one deformable group: host check)
    const int Cg = a.C / a.groups;
    float acc[KMAX], bsum = 0.f;
#pragma unroll
    for (int t = 0; t < KMAX; ++t) acc[t] = 0.f;
    const int r0 = blockIdx.x * rows_per_split, r1 = min(r0 + rows_per_split, nrows);
    for (int rb = r0; rb < r1; rb += GRP_PB) {
        const int npx = min(GRP_PB, r1 - rb);
        __syncthreads();   // (the previous chunk's readers are done)
        for (int e = tid; e < GRP_PB * NCO; e += 256) {
            const int px = e / NCO, c = e - px * NCO;
            float v = 0.f;
            if (px < npx) {
                const Lvl &L = find_level_by_row(a, rb + px);
                v = L.gout[(size_t)(rb + px - L.prow0) * a.opitch + cout0 + c];
            }
            gout_s[e] = v;
        }
        for (int e = tid; e < GRP_PB * K * Q4; e += 256) {
            const int c4 = e % Q4, rest = e / Q4;
            const int px = rest % GRP_PB, k = rest / GRP_PB;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (px < npx) {
                const Lvl &L = find_level_by_row(a, rb + px);
                const Tap tp = grp_tap(a, L, rb + px, k, dgi);
                if (tp.flags) {
                    float b00, b01, b10, b11;
                    corner_weights(tp, b00, b01, b10, b11);
                    b00 *= tp.m, b01 *= tp.m, b10 *= tp.m, b11 *= tp.m;
                    const float *xb = L.x + cin0 + 4 * c4;
                    const float4 x00 = *reinterpret_cast<const float4 *>(xb + tp.i00), x01 = *reinterpret_cast<const float4 *>(xb + tp.i01);
                    const float4 x10 = *reinterpret_cast<const float4 *>(xb + tp.i10), x11 = *reinterpret_cast<const float4 *>(xb + tp.i11);
                    v.x = b00 * x00.x + b01 * x01.x + b10 * x10.x + b11 * x11.x;
                    v.y = b00 * x00.y + b01 * x01.y + b10 * x10.y + b11 * x11.y;
                    v.z = b00 * x00.z + b01 * x01.z + b10 * x10.z + b11 * x11.z;
                    v.w = b00 * x00.w + b01 * x01.w + b10 * x10.w + b11 * x11.w;
                }
            }
            *reinterpret_cast<float4 *>(col_s + ((size_t)px * K + k) * NCH + 4 * c4) = v;
        }
        __syncthreads();
        for (int px = 0; px < npx; ++px) {
            const float gv = gout_s[px * NCO + lco];
            bsum += gv;
            const float *cp = col_s + (size_t)px * K * NCH + lci;
#pragma unroll
            for (int t = 0; t < KMAX; ++t)
                if (t < K) acc[t] = fmaf(gv, cp[t * NCH], acc[t]);
        }
    }
    const size_t nW = (size_t)a.Co * K * Cg;
    float *pw = part + (size_t)blockIdx.x * nW + ((size_t)(g * CG + co) * K) * Cg + ci;
#pragma unroll
    for (int t = 0; t < KMAX; ++t)
        if (t < K) pw[(size_t)t * Cg] = acc[t];
    if (part_b != nullptr && ci == 0) part_b[(size_t)blockIdx.x * a.Co + g * CG + co] = bsum;
}

// ---- forward ----
// out[p][g CG + co] = bias + sum_{k, ci} col[p][k][g CG + ci] * w[g CG + co][k][ci]        (deform_conv_cuda.cpp:662-684 per group)
// Workgroup = 32 pixels x 64 channels (NG = 64 / CG groups; CG = 8: two groups share a 16-wide matrix tile through a
// block-diagonal weight fragment), four waves; a wave owns 16 of the 64 output channels for both 16-pixel halves.  Per tap:
// the 32 x 64 sampled values are blended into LDS (rows padded to 68 floats: the A-fragment reads -- lane = pixel (l & 15),
// channel 4 step + (l >> 4) -- hit 64 different banks), the wave's weight fragments of the tap come straight from L2, then
// CG / 4 v_mfma_f32_16x16x4_f32 per pixel half.  Two LDS stages: tap k + 1 is gathered while tap k is multiplied.
constexpr int GF_PX = 32, GF_CH = 64, GF_ROW = 68;

__host__ __device__ inline size_t dcn_fwd_grouped_lds_bytes(int KD) { return (size_t)2 * GF_PX * GF_ROW * 4 + (size_t)GF_PX * KD * sizeof(Tap); }

template <int CG>
__global__ __launch_bounds__(256) void dcn_fwd_grouped_kernel(const DcnArgs a)
{
    static_assert(CG == 8 || CG == 16 || CG == 32, "channels per group");
    constexpr int KS = (CG < 16 ? 16 : CG) / 4;        // k-steps (4 input channels each) of a wave's reduction
    extern __shared__ __align__(16) float fsm[];
    float *stage = fsm;                                                 // [2][GF_PX][GF_ROW]
    Tap *tab = reinterpret_cast<Tap *>(fsm + 2 * GF_PX * GF_ROW);       // [GF_PX][K] (this block's deformable group)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.kh * a.kw;
    const int nspan = a.C / GF_CH;
    const int ptile = blockIdx.x / nspan, span = blockIdx.x - ptile * nspan;
    const Lvl &L = find_level(a, ptile);
    const int tile_p = (ptile - L.tile0) * GF_PX;
    const int c0 = span * GF_CH;                      // first (input and output) channel of the block
    const int dgi = c0 / (a.C / a.dg);
    const int Cg = a.C / a.groups;                    // == CG
    for (int e = tid; e < GF_PX * K; e += 256) {
        const int pl = e / K, k = e - pl * K;
        tab[e] = make_tap(a, L, tile_p + pl, k, dgi);
    }
    // staging role: thread = (pixel tid >> 3, float4 slots (tid & 7) and (tid & 7) + 8 of the 64 channels)
    const int spx = tid >> 3, sc4 = tid & 7;
    // wave's output channels: c0 + 16 wave .. + 15; its input channels: the group(s) those belong to
    const int wco0 = c0 + 16 * wave;
    const int wci0 = CG >= 16 ? (wco0 / CG) * CG : wco0;              // CG = 8: the two groups of the tile, 16 channels
    const int n = lane & 15, kk = lane >> 4;
    // B fragment of (tap, step): B[kk][n] = w[wco0 + n][tap][ci], ci = input channel (wci0 + 4 step + kk) relative to n's group
    const int gco = wco0 + n;                                          // this lane's output channel
    const int gbase = (gco / CG) * CG;                                 // first input channel of its group
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    __syncthreads();
    float4 xv[2][4];
    float bw[4];
    auto issue = [&](int k) {
        const Tap tp = tab[spx * K + k];
        float b00, b01, b10, b11;
        corner_weights(tp, b00, b01, b10, b11);
        bw[0] = b00 * tp.m, bw[1] = b01 * tp.m, bw[2] = b10 * tp.m, bw[3] = b11 * tp.m;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float *xb = L.x + c0 + 4 * (sc4 + 8 * h);
            xv[h][0] = *reinterpret_cast<const float4 *>(xb + tp.i00);
            xv[h][1] = *reinterpret_cast<const float4 *>(xb + tp.i01);
            xv[h][2] = *reinterpret_cast<const float4 *>(xb + tp.i10);
            xv[h][3] = *reinterpret_cast<const float4 *>(xb + tp.i11);
        }
    };
    auto commit = [&](float *buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float4 v;
            v.x = bw[0] * xv[h][0].x + bw[1] * xv[h][1].x + bw[2] * xv[h][2].x + bw[3] * xv[h][3].x;
            v.y = bw[0] * xv[h][0].y + bw[1] * xv[h][1].y + bw[2] * xv[h][2].y + bw[3] * xv[h][3].y;
            v.z = bw[0] * xv[h][0].z + bw[1] * xv[h][1].z + bw[2] * xv[h][2].z + bw[3] * xv[h][3].z;
            v.w = bw[0] * xv[h][0].w + bw[1] * xv[h][1].w + bw[2] * xv[h][2].w + bw[3] * xv[h][3].w;
            *reinterpret_cast<float4 *>(buf + spx * GF_ROW + 4 * (sc4 + 8 * h)) = v;
        }
    };
    float wf[KS];
    auto load_w = [&](int k) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int cin = wci0 + 4 * s + kk;                         // absolute input channel of this fragment element
            const int rel = cin - gbase;                               // ... inside lane n's group?  (CG = 8: block-diagonal)
            wf[s] = (rel >= 0 && rel < CG) ? a.w[((size_t)gco * K + k) * Cg + rel] : 0.f;
        }
    };
    issue(0);
    commit(stage);
    if (K > 1) issue(1);
    load_w(0);
    __syncthreads();
    for (int k = 0; k < K; ++k) {
        const float *bc = stage + (k & 1) * GF_PX * GF_ROW;
        float *bn = stage + ((k & 1) ^ 1) * GF_PX * GF_ROW;
        float af[2][KS];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int s = 0; s < KS; ++s) af[h][s] = bc[(16 * h + n) * GF_ROW + (wci0 - c0) + 4 * s + kk];
        if (k + 1 < K) commit(bn);            // registers hold the corners of tap k + 1
        if (k + 2 < K) issue(k + 2);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            acc[0] = mfma16(af[0][s], wf[s], acc[0]);
            acc[1] = mfma16(af[1][s], wf[s], acc[1]);
        }
        if (k + 1 < K) load_w(k + 1);
        __syncthreads();
    }
    // D[m][n]: lane holds rows m = 4 (lane >> 4) + i, column n = lane & 15
    const float bv = a.bias ? a.bias[gco] : 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pix = tile_p + 16 * h + 4 * kk + i;
            if (pix < L.P) L.out[(size_t)pix * a.opitch + gco] = acc[h][i] + bv;
        }
}

// ---- column gradients of the backward-data pass on the fp32 matrix instructions ----
//   gcol[(row, k)][c] = sum_{j < Co / groups} gout[row][g Cog + j] * w[g Cog + j][k][c - g Cg],   g = c / Cg      (UNWEIGHTED)
// -- what dcn_gcol_grouped_kernel (dcn_gather_kernels.h) computes with one fmaf chain per thread: 14.6 of the 108 ms of the config-4
// step, and ~5 ms per tower launch when the exact mode sends dense calls through it (round 6).  Here: workgroup = 32 launch-wide
// pixel rows x 64 columns, wave = 16 columns for both 16-row halves.  The grad_output tile is staged in LDS once and a wave keeps
// its A fragments (rows x the gout channels its columns reduce over: the group's Cog, or all Co of a dense call; below 16 channels
// per group the groups of a 16-column tile share the fragment through a block-diagonal weight fragment) in registers for all
// taps; per tap the weight fragments come from L2, KS steps of v_mfma_f32_16x16x4_f32 per half, then the 16 x 16 tile of the tap
// is stored.  Exact fp32 (the fmaf chain of the reference up to summation order).
// KS = k-steps of four gout channels: 4 (groups of 4 / 8 / 16 channels), 8 (32), Co / 4 for a dense call (Co <= 256).
constexpr int GC_PX = 32, GC_COLS = 64;
__host__ __device__ inline int dcn_gcol_mfma_pitch(int nred) { return nred + 4; }   // LDS row pitch in floats: 4 n + kk hits 64 banks
__host__ __device__ inline size_t dcn_gcol_mfma_lds_bytes(int nred) { return (size_t)GC_PX * dcn_gcol_mfma_pitch(nred) * sizeof(float); }

template <int KS>
__global__ __launch_bounds__(256) void dcn_gcol_mfma_kernel(const DcnArgs a, int nred)
{
    extern __shared__ __align__(16) float csm[];   // [GC_PX][pitch]: the gout channels [red0, red0 + nred) of the tile's rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.kh * a.kw, C = a.C;
    const int Cg = C / a.groups, Cog = a.Co / a.groups;
    const int nspan = C / GC_COLS;
    const int ptile = blockIdx.x / nspan, span = blockIdx.x - ptile * nspan;
    const Lvl &L = find_level(a, ptile);
    const int p0 = (ptile - L.tile0) * GC_PX;
    const int c0 = span * GC_COLS;
    // gout channels this workgroup reduces over: all of them (dense), or those of the span's groups
    const int red0 = a.groups == 1 ? 0 : (c0 / Cg) * Cog;
    const int pitch = dcn_gcol_mfma_pitch(nred);
    for (int e = tid; e < GC_PX * (nred / 4); e += 256) {
        const int px = e / (nred / 4), q = e - px * (nred / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p0 + px < L.P) v = *reinterpret_cast<const float4 *>(L.gout + (size_t)(p0 + px) * a.opitch + red0 + 4 * q);
        *reinterpret_cast<float4 *>(csm + px * pitch + 4 * q) = v;
    }
    __syncthreads();
    const int n = lane & 15, kk = lane >> 4;
    const int col = c0 + 16 * wave + n;                 // this lane's output column (B / D column n)
    const int g = col / Cg, ci = col - g * Cg;
    // the wave's reduction window inside the staged channels: KS * 4 channels starting at wred0 (relative to red0)
    const int wred0 = a.groups == 1 ? 0 : (c0 + 16 * wave) / Cg * Cog - red0;
    float af[2][KS];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int s = 0; s < KS; ++s) af[h][s] = csm[(16 * h + n) * pitch + wred0 + 4 * s + kk];
    const size_t row0 = (size_t)(L.prow0 + p0);
    // weight fragment element (tap k, step s): w[j][k][ci], j = red0 + wred0 + 4 s + kk -- a raw buffer load at a 32-bit offset
    // (per-lane part: j of step 0 and ci; per-step and per-tap parts are wave-uniform), zero for the other groups of the tile
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.w), 0, (int)((size_t)a.Co * K * Cg * 4), 0x00020000);
    const int wv0 = (((red0 + wred0 + kk) * K) * Cg + ci) * 4, wstep = 4 * K * Cg * 4;
    const int jrel0 = red0 + wred0 + kk - g * Cog;          // gout channel of step 0 relative to the lane's group
    for (int k = 0; k < K; ++k) {
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        constexpr int WB = KS < 16 ? KS : 16;   // weight fragments in flight (a dense call has up to 64 k-steps per tap)
#pragma unroll
        for (int s0 = 0; s0 < KS; s0 += WB) {
            float wf[WB];
#pragma unroll
            for (int s = 0; s < WB; ++s) {
                const float v = buf_load_f32(wrs, wv0 + (s0 + s) * wstep, k * Cg * 4);
                wf[s] = (unsigned)(jrel0 + 4 * (s0 + s)) < (unsigned)Cog ? v : 0.f;
            }
#pragma unroll
            for (int s = 0; s < WB; ++s) {
                acc[0] = mfma16(af[0][s0 + s], wf[s], acc[0]);
                acc[1] = mfma16(af[1][s0 + s], wf[s], acc[1]);
            }
            if (KS > WB) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int px = 16 * h + 4 * kk + i;
                if (p0 + px < L.P) a.gcol[((row0 + px) * K + k) * C + col] = acc[h][i];
            }
    }
}

}  // namespace lsn
