// Deformable-convolution family for gfx950: DCNv1 / DCNv2 (modulated) / pyramid DCN, forward,
// backward-data (grad input / offset / mask) and backward-weight, as fused gather + MFMA
// implicit GEMMs.  Nothing like the reference's C*9*H*W column buffer ever reaches HBM
// (reference: im2col kernel + cuBLAS addmm_, deform_conv_cuda.cpp:662-684).
//
// Data layout (HBM):  activations NHWC  x[b][y][x][c];  weight OHWI  w[co][k*Cg + ci]
// (k = i*kw + j the tap, ci the channel inside the conv group);  offsets / masks via strides.
//
// GEMM view, per conv group g:   out[p][co] = sum_{k,ci} col[p][k,ci] * w[co][k,ci]
//   p  = output pixel (b,ho,wo), flattened over ALL levels of a batched launch,
//   col[p][k,ci] = mask * bilinear(x[b,:,:,g*Cg+ci], sample position of tap k at p).
// The K dimension is walked in chunks of one tap x 32 channels; the channel chunking never
// crosses a conv-group or deformable-group boundary, so every chunk has one (g, dgi).
//
// fp32 MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4_f32) runs at the fp32 vector rate (157 TF peak),
// i.e. 64 / 32 cycles per instruction: the matrix pipe, not LDS or L2 bandwidth, is the bound,
// so tiles are kept small enough to balance 700..1400 tiles over 256 CUs and the gather/stage
// work of chunk t+1 is issued before the MFMA phase of chunk t.
#pragma once
#include "common.h"

namespace lsn {

constexpr int MAXLV = 16;

struct Lvl {
    const float *x, *off, *msk, *gout;
    float *out, *gx, *goff, *gmsk;
    int B, H, W, Ho, Wo;
    int P;      // B*Ho*Wo
    int tile0;  // first tile (fwd: 64-px tiles, bwd-data: 64-px tiles, wgrad: 32-px steps)
    float sh, sw;
    int osb, osc, osh, osw;  // offset (and grad_offset) element strides
    int msb, msc, msh, msw;  // mask (and grad_mask) element strides
};

struct DcnArgs {
    Lvl lv[MAXLV];
    int nlv, ntiles;
    const float *w, *bias;
    float *gw, *gb;
    int C, Co, kh, kw, stride, pad, dil, groups, dg;
    int SL;   // channel segment length = min(C/groups, C/dg): constant (g, dgi) inside a segment
    int msig; // mask tensor holds logits: apply sigmoid on read, chain it into grad_mask
};

// One sampling position: the four clamped NHWC element offsets of its bilinear corners (channel 0),
// the fractional parts, the modulation scalar and a validity bit per corner.
struct __align__(16) Tap {
    int i00, i01, i10, i11;
    float ly, lx, m;
    int flags;
};

__device__ __forceinline__ const Lvl &find_level(const DcnArgs &a, int tile)
{
    int li = 0;
    while (li + 1 < a.nlv && tile >= a.lv[li + 1].tile0) ++li;
    return a.lv[li];
}

// Sample position exactly as the oracle / reference compute it:
//   py = float(ho*stride - pad + i*dil) * scale_h + dy     (kernel.cu:227-228, 281-282, 892-893)
__device__ __forceinline__ Tap make_tap(const DcnArgs &a, const Lvl &L, int pix, int k, int dgi)
{
    Tap t;
    t.i00 = t.i01 = t.i10 = t.i11 = 0;
    t.ly = t.lx = t.m = 0.f;
    t.flags = 0;
    if (pix >= L.P) return t;
    const int K = a.kh * a.kw;
    const int HWo = L.Ho * L.Wo;
    const int b = pix / HWo;
    const int rem = pix - b * HWo;
    const int ho = rem / L.Wo, wo = rem - ho * L.Wo;
    const int i = k / a.kw, j = k - i * a.kw;
    const float *op = L.off + (size_t)b * L.osb + (size_t)ho * L.osh + (size_t)wo * L.osw;
    const float oy = op[(size_t)(dgi * 2 * K + 2 * k) * L.osc];
    const float ox = op[(size_t)(dgi * 2 * K + 2 * k + 1) * L.osc];
    float m = L.msk ? L.msk[(size_t)b * L.msb + (size_t)(dgi * K + k) * L.msc + (size_t)ho * L.msh +
                            (size_t)wo * L.msw]
                    : 1.f;
    if (a.msig && L.msk) m = 1.f / (1.f + expf(-m));
    const float py = __fadd_rn(__fmul_rn((float)(ho * a.stride - a.pad + i * a.dil), L.sh), oy);
    const float px = __fadd_rn(__fmul_rn((float)(wo * a.stride - a.pad + j * a.dil), L.sw), ox);
    if (py > -1.f && px > -1.f && py < (float)L.H && px < (float)L.W) {
        const float fy = floorf(py), fx = floorf(px);
        const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
        t.ly = py - fy;
        t.lx = px - fx;
        t.m = m;
        const bool vy0 = y0 >= 0, vy1 = y1 <= L.H - 1, vx0 = x0 >= 0, vx1 = x1 <= L.W - 1;
        const int cy0 = max(y0, 0), cy1 = min(y1, L.H - 1), cx0 = max(x0, 0), cx1 = min(x1, L.W - 1);
        const int r0 = (b * L.H + cy0) * L.W, r1 = (b * L.H + cy1) * L.W;
        t.i00 = (r0 + cx0) * a.C;
        t.i01 = (r0 + cx1) * a.C;
        t.i10 = (r1 + cx0) * a.C;
        t.i11 = (r1 + cx1) * a.C;
        t.flags = (int)(vy0 && vx0) | ((int)(vy0 && vx1) << 1) | ((int)(vy1 && vx0) << 2) |
                  ((int)(vy1 && vx1) << 3);
    }
    return t;
}

__device__ __forceinline__ void corner_weights(const Tap &t, float &b00, float &b01, float &b10, float &b11)
{
    const float hy = 1.f - t.ly, hx = 1.f - t.lx;
    b00 = (t.flags & 1) ? hy * hx : 0.f;
    b01 = (t.flags & 2) ? hy * t.lx : 0.f;
    b10 = (t.flags & 4) ? t.ly * hx : 0.f;
    b11 = (t.flags & 8) ? t.ly * t.lx : 0.f;
}

// chunk index -> (tap k, channel offset inside the conv group, #valid channels, deformable group)
struct Chunk {
    int k, c0, nval, dgi;
};
template <int CK>
__device__ __forceinline__ Chunk decode_chunk(const DcnArgs &a, int g, int t, int segs, int ncc)
{
    // order: tap-major, then segment, then CK-channel sub-chunk
    Chunk c;
    c.k = t / (segs * ncc);
    const int r = t - c.k * segs * ncc;
    const int seg = r / ncc, cc = r - seg * ncc;
    c.c0 = seg * a.SL + cc * CK;
    c.nval = min(CK, a.SL - cc * CK);
    const int Cg = a.C / a.groups;
    c.dgi = (g * Cg + seg * a.SL) / (a.C / a.dg);
    return c;
}

// =============================================================================================
// Forward:  block = BM output pixels x BN output channels of one conv group; 4 waves (WM x WN),
// each owning TM x TN accumulator tiles of 32x32 (v_mfma_f32_32x32x2_f32).
//   LDS:  As[BM][33]  sampled+modulated values of the current chunk (pixel rows, k contiguous)
//         Bs[BN][33]  weight chunk (co rows, k contiguous);  stride 33 -> conflict-free both ways
//         tab[BM][K*dg] sampling table built once per block
// =============================================================================================
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void dcn_fwd_kernel(const DcnArgs a)
{
    constexpr int BK = 32, LDK = BK + 1;
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    static_assert(WM * WN == 4, "4 waves per block");
    static_assert(TM >= 1 && TN >= 1, "tile too small");
    extern __shared__ __align__(16) unsigned char smem[];
    float *As = reinterpret_cast<float *>(smem);
    float *Bs = As + BM * LDK;
    Tap *tab = reinterpret_cast<Tap *>(Bs + BN * LDK);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int K = a.kh * a.kw, KD = K * a.dg;
    const int Cg = a.C / a.groups, Cog = a.Co / a.groups, Kdim = K * Cg;

    const Lvl &L = find_level(a, blockIdx.x);
    const int tile_p = (blockIdx.x - L.tile0) * BM;
    const int g = blockIdx.z;
    const int co_blk = blockIdx.y * BN;
    const int nco = min(BN, Cog - co_blk);
    const int co_base = g * Cog + co_blk;

    for (int e = tid; e < BM * KD; e += 256) {
        const int pl = e / KD, r = e - pl * KD;
        const int dgi = r / K, k = r - dgi * K;
        tab[e] = make_tap(a, L, tile_p + pl, k, dgi);
    }

    const int segs = Cg / a.SL, ncc = (a.SL + BK - 1) / BK;
    const int T = K * segs * ncc;

    const int kk = tid & 31, prow = tid >> 5;  // gather: channel lane, pixel row (8 rows per pass)
    constexpr int NPA = BM / 8;
    const int wq = tid & 7, wrow = tid >> 3;   // weights: float4 slot along k, co row (32 rows per pass)
    constexpr int NPB = BN / 32;
    const bool vec4 = (Cg & 3) == 0;
    float xv[NPA][4];
    float4 wv[NPB];

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto load_chunk = [&](int t) {
        const Chunk ch = decode_chunk<BK>(a, g, t, segs, ncc);
        const int c = g * Cg + ch.c0 + (kk < ch.nval ? kk : 0);
#pragma unroll
        for (int ps = 0; ps < NPA; ++ps) {
            const Tap *tp = &tab[(ps * 8 + prow) * KD + ch.dgi * K + ch.k];
            const int4 idx = *reinterpret_cast<const int4 *>(tp);
            xv[ps][0] = L.x[idx.x + c];
            xv[ps][1] = L.x[idx.y + c];
            xv[ps][2] = L.x[idx.z + c];
            xv[ps][3] = L.x[idx.w + c];
        }
#pragma unroll
        for (int ps = 0; ps < NPB; ++ps) {
            const int col = ps * 32 + wrow;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int rem = ch.nval - wq * 4;
            if (col < nco && rem > 0) {
                const float *wp = a.w + (size_t)(co_base + col) * Kdim + ch.k * Cg + ch.c0 + wq * 4;
                if (vec4 && rem >= 4) {
                    v = *reinterpret_cast<const float4 *>(wp);
                } else {
                    v.x = wp[0];
                    if (rem > 1) v.y = wp[1];
                    if (rem > 2) v.z = wp[2];
                    if (rem > 3) v.w = wp[3];
                }
            }
            wv[ps] = v;
        }
    };

    auto store_chunk = [&](int t) {
        const Chunk ch = decode_chunk<BK>(a, g, t, segs, ncc);
        const bool cval = kk < ch.nval;
#pragma unroll
        for (int ps = 0; ps < NPA; ++ps) {
            const Tap tp = tab[(ps * 8 + prow) * KD + ch.dgi * K + ch.k];
            float b00, b01, b10, b11;
            corner_weights(tp, b00, b01, b10, b11);
            float v = b00 * xv[ps][0] + b01 * xv[ps][1] + b10 * xv[ps][2] + b11 * xv[ps][3];
            v *= tp.m;
            As[(ps * 8 + prow) * LDK + kk] = cval ? v : 0.f;
        }
#pragma unroll
        for (int ps = 0; ps < NPB; ++ps) {
            float *bp = Bs + (ps * 32 + wrow) * LDK + wq * 4;
            bp[0] = wv[ps].x;
            bp[1] = wv[ps].y;
            bp[2] = wv[ps].z;
            bp[3] = wv[ps].w;
        }
    };

    __syncthreads();  // sampling table complete
    load_chunk(0);
    for (int t = 0; t < T; ++t) {
        store_chunk(t);
        __syncthreads();
        if (t + 1 < T) load_chunk(t + 1);  // in flight during the MFMA phase below
        const float *ap = As + (wm * TM * 32 + (lane & 31)) * LDK + (lane >> 5);
        const float *bp = Bs + (wn * TN * 32 + (lane & 31)) * LDK + (lane >> 5);
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            float af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = ap[i * 32 * LDK + 2 * s];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = bp[j * 32 * LDK + 2 * s];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(af[i], bf[j], acc[i][j]);
        }
        __syncthreads();
    }

    // epilogue: D rows = pixels, D cols = output channels (32 consecutive floats per half-wave)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = wn * TN * 32 + j * 32 + (lane & 31);
            if (col >= nco) continue;
            const float bv = a.bias ? a.bias[co_base + col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pix = tile_p + wm * TM * 32 + i * 32 + mfma32_row(r, lane);
                if (pix < L.P) L.out[(size_t)pix * a.Co + co_base + col] = acc[i][j][r] + bv;
            }
        }
}

// =============================================================================================
// Backward-data:  gcol[p][k,ci] = sum_co gout[p][co] * w[co][k,ci]   (never stored), then
//   grad_input  += bilinear-weighted scatter of gcol*mask            (fp32 atomics, kernel.cu:913-970)
//   grad_offset  = sum_ci gcol*mask * d(bilinear)/d(py,px)           (kernel.cu:973-1044)
//   grad_mask    = sum_ci gcol * bilinear
// Block = 64 pixels, 4 waves x 16 pixel rows (v_mfma_f32_16x16x4_f32).  The A operand (the
// wave's 16 gout rows, RED <= 256 output channels of one group) lives in registers for the
// whole block; the weight chunk [RED][32 k-columns] is streamed through LDS (XOR-swizzled so
// the 4 k-quarters of a B read hit disjoint banks).  goff/gmask partial sums are reduced over
// the 16 channel lanes per (tap, segment) and accumulated in LDS (rows are wave-private).
// =============================================================================================
constexpr int BWD_BM = 64;
template <int RED>
__global__ __launch_bounds__(256, 2) void dcn_bwd_data_kernel(const DcnArgs a)
{
    constexpr int BK = 32, QR = RED / 4;  // QR reduction indices per lane quarter
    extern __shared__ __align__(16) unsigned char smem[];
    float *Bs = reinterpret_cast<float *>(smem);               // [RED][32] swizzled
    Tap *tab = reinterpret_cast<Tap *>(Bs + RED * BK);         // [64][K*dg]
    const int K = a.kh * a.kw, KD = K * a.dg;
    float *gacc = reinterpret_cast<float *>(tab + BWD_BM * KD);  // [64][KD][3]  (dy, dx, mask)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j16 = lane & 15, kq = lane >> 4;
    const int Cg = a.C / a.groups, Cog = a.Co / a.groups, Kdim = K * Cg;

    const Lvl &L = find_level(a, blockIdx.x);
    const int tile_p = (blockIdx.x - L.tile0) * BWD_BM;

    for (int e = tid; e < BWD_BM * KD; e += 256) {
        const int pl = e / KD, r = e - pl * KD;
        const int dgi = r / K, k = r - dgi * K;
        tab[e] = make_tap(a, L, tile_p + pl, k, dgi);
    }
    for (int e = tid; e < BWD_BM * KD * 3; e += 256) gacc[e] = 0.f;

    const int segs = Cg / a.SL, ncc = (a.SL + BK - 1) / BK;
    const int T = K * segs * ncc;
    const int nrb = (Cog + RED - 1) / RED;  // reduction blocks (1 for Co/groups <= RED)
    const bool want_off = (L.goff != nullptr) || (L.gmsk != nullptr);

    const int wq = tid & 7, wrow = tid >> 3;  // weight staging: float4 slot along k, 32 rows/pass
    constexpr int NPB = RED / 32;
    const bool vec4 = (Cg & 3) == 0;
    const bool avec4 = (a.Co & 3) == 0 && (Cog & 3) == 0;

    const int my_pix = tile_p + wave * 16 + j16;  // A operand row (pixel) of this lane

    float areg[QR];
    float4 wv[NPB];
    __syncthreads();

    // A operand: gout[pix][g*Cog + rb*RED + kq*QR + s], s = 0..QR-1 (this lane's k-quarter)
    auto load_a = [&](int g, int rb) {
        const int cb = rb * RED + kq * QR;
#pragma unroll
        for (int s4 = 0; s4 < QR / 4; ++s4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int rem = Cog - (cb + s4 * 4);
            if (my_pix < L.P && rem > 0) {
                const float *gp = L.gout + (size_t)my_pix * a.Co + g * Cog + cb + s4 * 4;
                if (avec4 && rem >= 4) {
                    v = *reinterpret_cast<const float4 *>(gp);
                } else {
                    v.x = gp[0];
                    if (rem > 1) v.y = gp[1];
                    if (rem > 2) v.z = gp[2];
                    if (rem > 3) v.w = gp[3];
                }
            }
            areg[s4 * 4 + 0] = v.x;
            areg[s4 * 4 + 1] = v.y;
            areg[s4 * 4 + 2] = v.z;
            areg[s4 * 4 + 3] = v.w;
        }
    };
    // weight slab of (chunk t, reduction block rb): rows = output channels, 32 k-columns
    auto load_w = [&](int g, int t, int rb) {
        const Chunk ch = decode_chunk<BK>(a, g, t, segs, ncc);
#pragma unroll
        for (int ps = 0; ps < NPB; ++ps) {
            const int row = ps * 32 + wrow;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int rem = ch.nval - wq * 4;
            if (rb * RED + row < Cog && rem > 0) {
                const float *wp = a.w + (size_t)(g * Cog + rb * RED + row) * Kdim + ch.k * Cg + ch.c0 + wq * 4;
                if (vec4 && rem >= 4) {
                    v = *reinterpret_cast<const float4 *>(wp);
                } else {
                    v.x = wp[0];
                    if (rem > 1) v.y = wp[1];
                    if (rem > 2) v.z = wp[2];
                    if (rem > 3) v.w = wp[3];
                }
            }
            wv[ps] = v;
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int ps = 0; ps < NPB; ++ps) {
            const int row = ps * 32 + wrow;
            const int sw = ((row / QR) & 1) << 4;
            *reinterpret_cast<float4 *>(Bs + row * BK + ((wq * 4) ^ sw)) = wv[ps];
        }
    };

    for (int g = 0; g < a.groups; ++g) {
        if (nrb == 1) load_a(g, 0);
        load_w(g, 0, 0);
        for (int t = 0; t < T; ++t) {
            const Chunk ch = decode_chunk<BK>(a, g, t, segs, ncc);
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            for (int rb = 0; rb < nrb; ++rb) {
                if (nrb > 1) load_a(g, rb);  // Co/groups > RED: re-read the gout rows per slab
                store_w();
                __syncthreads();
                if (rb + 1 < nrb)
                    load_w(g, t, rb + 1);
                else if (t + 1 < T)
                    load_w(g, t + 1, 0);
                {
                    const int sw = (kq & 1) << 4;
                    const float *bp = Bs + (kq * QR) * BK;
                    const int c0i = j16 ^ sw, c1i = (16 + j16) ^ sw;
#pragma unroll
                    for (int s = 0; s < QR; ++s) {
                        const float b0 = bp[s * BK + c0i];
                        const float b1 = bp[s * BK + c1i];
                        acc0 = mfma16(areg[s], b0, acc0);
                        acc1 = mfma16(areg[s], b1, acc1);
                    }
                }
                if (rb + 1 < nrb) __syncthreads();  // slab consumed; next slab may overwrite Bs
            }

            // ---- consume gcol[16 px][32 ch] of this wave: D row = 4*kq + r, col = tn*16 + j16 ----
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const int cl = tn * 16 + j16;
                const bool cval = cl < ch.nval;
                const int c = g * Cg + ch.c0 + (cval ? cl : 0);
                float xv[4][4];
                if (want_off) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const Tap *tp = &tab[(wave * 16 + kq * 4 + r) * KD + ch.dgi * K + ch.k];
                        const int4 idx = *reinterpret_cast<const int4 *>(tp);
                        xv[r][0] = L.x[idx.x + c];
                        xv[r][1] = L.x[idx.y + c];
                        xv[r][2] = L.x[idx.z + c];
                        xv[r][3] = L.x[idx.w + c];
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float gval = cval ? (tn == 0 ? acc0[r] : acc1[r]) : 0.f;
                    const Tap tp = tab[(wave * 16 + kq * 4 + r) * KD + ch.dgi * K + ch.k];
                    float b00, b01, b10, b11;
                    corner_weights(tp, b00, b01, b10, b11);
                    const float gm = gval * tp.m;
                    if (L.gx != nullptr && cval) {
                        if (b00 != 0.f) atomic_add_f32(L.gx + tp.i00 + c, b00 * gm);
                        if (b01 != 0.f) atomic_add_f32(L.gx + tp.i01 + c, b01 * gm);
                        if (b10 != 0.f) atomic_add_f32(L.gx + tp.i10 + c, b10 * gm);
                        if (b11 != 0.f) atomic_add_f32(L.gx + tp.i11 + c, b11 * gm);
                    }
                    if (want_off) {
                        const float hy = 1.f - tp.ly, hx = 1.f - tp.lx;
                        const float v00 = (tp.flags & 1) ? xv[r][0] : 0.f;
                        const float v01 = (tp.flags & 2) ? xv[r][1] : 0.f;
                        const float v10 = (tp.flags & 4) ? xv[r][2] : 0.f;
                        const float v11 = (tp.flags & 8) ? xv[r][3] : 0.f;
                        // coordinate weights, kernel.cu:145-188 / 800-845
                        const float dy = hx * (v10 - v00) + tp.lx * (v11 - v01);
                        const float dx = hy * (v01 - v00) + tp.ly * (v11 - v10);
                        const float bil =
                            hy * hx * v00 + hy * tp.lx * v01 + tp.ly * hx * v10 + tp.ly * tp.lx * v11;
                        // reduce over the 16 channel lanes of this k-quarter, park in LDS (row is
                        // private to this wave, so a plain read-modify-write is race-free)
                        float vy = gm * dy, vx = gm * dx, vm = gval * bil;
#pragma unroll
                        for (int o = 8; o >= 1; o >>= 1) {
                            vy += __shfl_xor(vy, o);
                            vx += __shfl_xor(vx, o);
                            vm += __shfl_xor(vm, o);
                        }
                        if (j16 == 0) {
                            float *ga = gacc + ((wave * 16 + kq * 4 + r) * KD + ch.dgi * K + ch.k) * 3;
                            ga[0] += vy;
                            ga[1] += vx;
                            ga[2] += vm;
                        }
                    }
                }
            }
            __syncthreads();  // Bs free for the next chunk
        }
    }
    __syncthreads();

    // ---- write grad_offset / grad_mask for this tile ----
    for (int e = tid; e < BWD_BM * KD; e += 256) {
        const int pl = e / KD, r = e - pl * KD;
        const int dgi = r / K, k = r - dgi * K;
        const int pix = tile_p + pl;
        if (pix >= L.P) continue;
        const int HWo = L.Ho * L.Wo;
        const int b = pix / HWo, rem = pix - b * HWo;
        const int ho = rem / L.Wo, wo = rem - ho * L.Wo;
        const float *ga = gacc + e * 3;
        if (L.goff) {
            float *op = L.goff + (size_t)b * L.osb + (size_t)ho * L.osh + (size_t)wo * L.osw;
            op[(size_t)(dgi * 2 * K + 2 * k) * L.osc] = ga[0];
            op[(size_t)(dgi * 2 * K + 2 * k + 1) * L.osc] = ga[1];
        }
        if (L.gmsk) {
            float gm = ga[2];
            if (a.msig) {  // d sigmoid: m (1 - m); out-of-range samples have ga[2] == 0 already
                const float m = tab[e].m;
                gm *= m * (1.f - m);
            }
            L.gmsk[(size_t)b * L.msb + (size_t)(dgi * K + k) * L.msc + (size_t)ho * L.msh + (size_t)wo * L.msw] = gm;
        }
    }
}

// =============================================================================================
// Backward-weight:  gw[co][k,ci] += sum_p gout[p][co] * col[p][k,ci];  gb[co] += sum_p gout[p][co].
// GEMM with M = output channels (<=256 per block), N = 64 k-columns (one tap x 64 channels),
// reduction over pixels.  grid.x = column blocks, grid.y = pixel splits, grid.z = co blocks;
// each block walks its share of the 32-pixel steps of all levels, re-gathering col on the fly
// (the reference re-runs im2col for the same purpose, deform_conv_cuda.cpp:770-773), and ends
// with fp32 atomics into gw (zero-filled by the launcher).
// =============================================================================================
constexpr int WG_BP = 32, WG_BN = 64, WG_BM = 256;
__global__ __launch_bounds__(256, 2) void dcn_wgrad_kernel(const DcnArgs a, int nsteps)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float *As = reinterpret_cast<float *>(smem);          // [32 px][256 co]
    float *Bs = As + WG_BP * WG_BM;                        // [32 px][64 kcol]
    Tap *tab = reinterpret_cast<Tap *>(Bs + WG_BP * WG_BN);  // [2][32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.kh * a.kw;
    const int Cg = a.C / a.groups, Cog = a.Co / a.groups, Kdim = K * Cg;
    const int segs = Cg / a.SL, ncc = (a.SL + WG_BN - 1) / WG_BN;
    const int ncol_g = K * segs * ncc;  // column blocks per group
    const int g = blockIdx.x / ncol_g;
    const Chunk ch = decode_chunk<WG_BN>(a, g, blockIdx.x - g * ncol_g, segs, ncc);
    const int co_blk = blockIdx.z * WG_BM;
    const int nco = min(WG_BM, Cog - co_blk);
    const int co_base = g * Cog + co_blk;

    const int st_begin = (int)((long long)nsteps * blockIdx.y / gridDim.y);
    const int st_end = (int)((long long)nsteps * (blockIdx.y + 1) / gridDim.y);

    const int kk = tid & 63, prow = tid >> 6;  // gather: channel lane, 4 pixel rows per pass
    constexpr int NPA = WG_BP / 4;              // 8 passes
    const bool cval = kk < ch.nval;
    const int c = g * Cg + ch.c0 + (cval ? kk : 0);
    const int gq = tid & 63, grow = tid >> 6;  // gout: float4 slot along co, 4 rows per pass
    const bool avec4 = (a.Co & 3) == 0 && (Cog & 3) == 0;
    const bool do_bias = (a.gb != nullptr) && ch.k == 0 && ch.c0 == 0;

    float xv[NPA][4];
    float4 gv[NPA];
    float bias_acc = 0.f;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto build_tab = [&](int st, int buf) {
        if (tid < WG_BP) {
            const Lvl &L = find_level(a, st);
            tab[buf * WG_BP + tid] = make_tap(a, L, (st - L.tile0) * WG_BP + tid, ch.k, ch.dgi);
        }
    };
    auto load_step = [&](int st, int buf) {
        const Lvl &L = find_level(a, st);
        const int p0 = (st - L.tile0) * WG_BP;
#pragma unroll
        for (int ps = 0; ps < NPA; ++ps) {
            const Tap *tp = &tab[buf * WG_BP + ps * 4 + prow];
            const int4 idx = *reinterpret_cast<const int4 *>(tp);
            xv[ps][0] = L.x[idx.x + c];
            xv[ps][1] = L.x[idx.y + c];
            xv[ps][2] = L.x[idx.z + c];
            xv[ps][3] = L.x[idx.w + c];
        }
#pragma unroll
        for (int ps = 0; ps < NPA; ++ps) {
            const int pix = p0 + ps * 4 + grow;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int rem = nco - gq * 4;
            if (pix < L.P && rem > 0) {
                const float *gp = L.gout + (size_t)pix * a.Co + co_base + gq * 4;
                if (avec4 && rem >= 4) {
                    v = *reinterpret_cast<const float4 *>(gp);
                } else {
                    v.x = gp[0];
                    if (rem > 1) v.y = gp[1];
                    if (rem > 2) v.z = gp[2];
                    if (rem > 3) v.w = gp[3];
                }
            }
            gv[ps] = v;
        }
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int ps = 0; ps < NPA; ++ps) {
            const Tap tp = tab[buf * WG_BP + ps * 4 + prow];
            float b00, b01, b10, b11;
            corner_weights(tp, b00, b01, b10, b11);
            float v = b00 * xv[ps][0] + b01 * xv[ps][1] + b10 * xv[ps][2] + b11 * xv[ps][3];
            v *= tp.m;
            Bs[(ps * 4 + prow) * WG_BN + kk] = cval ? v : 0.f;
            *reinterpret_cast<float4 *>(As + (ps * 4 + grow) * WG_BM + gq * 4) = gv[ps];
        }
    };

    if (st_begin < st_end) {
        build_tab(st_begin, 0);
        __syncthreads();
        load_step(st_begin, 0);
        for (int st = st_begin; st < st_end; ++st) {
            const int buf = (st - st_begin) & 1;
            store_step(buf);
            if (st + 1 < st_end) build_tab(st + 1, buf ^ 1);
            __syncthreads();
            if (st + 1 < st_end) load_step(st + 1, buf ^ 1);
            if (do_bias) {
#pragma unroll
                for (int p = 0; p < WG_BP; ++p) bias_acc += As[p * WG_BM + tid];
            }
            const float *ap = As + (lane >> 5) * WG_BM + wave * 64 + (lane & 31);
            const float *bp = Bs + (lane >> 5) * WG_BN + (lane & 31);
#pragma unroll
            for (int s = 0; s < WG_BP / 2; ++s) {
                const float a0 = ap[2 * s * WG_BM], a1 = ap[2 * s * WG_BM + 32];
                const float b0 = bp[2 * s * WG_BN], b1 = bp[2 * s * WG_BN + 32];
                acc[0][0] = mfma32(a0, b0, acc[0][0]);
                acc[0][1] = mfma32(a0, b1, acc[0][1]);
                acc[1][0] = mfma32(a1, b0, acc[1][0]);
                acc[1][1] = mfma32(a1, b1, acc[1][1]);
            }
            __syncthreads();
        }
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = j * 32 + (lane & 31);
            if (col >= ch.nval) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wave * 64 + i * 32 + mfma32_row(r, lane);
                if (row < nco)
                    atomic_add_f32(a.gw + (size_t)(co_base + row) * Kdim + ch.k * Cg + ch.c0 + col, acc[i][j][r]);
            }
        }
    if (do_bias && tid < nco) atomic_add_f32(a.gb + co_base + tid, bias_acc);
}

}  // namespace lsn
