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
#include <type_traits>
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
    int prow0;  // first pixel row of this level in the launch-wide numbering (backward: column-gradient buffer)
    int abase;  // first anchor of the grad_input buffer this level scatters into (levels may share one)
};

struct DcnArgs {
    Lvl lv[MAXLV];
    int nlv, ntiles;
    const float *w, *bias;
    float *gw, *gb;
    int C, Co, kh, kw, stride, pad, dil, groups, dg;
    int SL;   // channel segment length = min(C/groups, C/dg): constant (g, dgi) inside a segment
    int msig; // mask tensor holds logits: apply sigmoid on read, chain it into grad_mask
    const unsigned short *wtp;   // pre-split bf16 weight planes (forward: [Co][K][Cg]; backward-data: [K][C][Co])
    float *gcol;   // backward-data: mask-weighted column gradients [(prow0 + pix) * K + k][C] for the gather pass
    const struct Tap *gtap;   // backward: sampling table of every (pixel, tap) of the launch, k-major:
    int wg_vec;               // weight gradient: tensors admit 8-byte buffer loads (bit 0: input, bit 1: grad_output)
    int gtap_rows;            // [kd * gtap_rows + prow0 + pix], gtap_rows = pixel rows of all levels (written once by
                              // dcn_bin_kernel; the backward kernels copy instead of recomputing)
    float *wg_part, *wg_part_b;   // weight gradient (dcn_wgrad_xn_kernel): per-pixel-split partial gradients [split][Co K Cg] and
                                  // [split][Co], added up by conv_wgrad_reduce_kernel in a fixed order; NULL: fp32 atomics into gw
    int mm;                   // a.wtp is in MFMA fragment order (conv_wfrag_kernel) for the kernels of dcn_mm_kernels.h
    int opitch;               // floats per pixel of out / gout (>= Co; the kernels of dcn_mm_kernels.h and conv_mm_rows only)
    int wtp_bytes;
    long long *dbg;  // optional phase timestamps of block `dbg_block`, wave 0 (lsn_debug_phase_clocks)
    int dbg_block;
};

// One sampling position: the four clamped NHWC element offsets of its bilinear corners (channel 0),
// the fractional parts, the modulation scalar and a validity bit per corner.
struct __align__(16) Tap {
    int i00, i01, i10, i11;
    float ly, lx, m;
    int flags;
};

// phase stamp: thread 0 of the chosen block appends the shader clock (s_memtime) to a.dbg
#define LSN_STAMP(slot)                                                                  \
    do {                                                                                 \
        if (a.dbg != nullptr && blockIdx.x == (unsigned)(a.dbg_block & 0xfffff) && threadIdx.x == 0 && \
            dbg_n < 512)                                                                 \
            a.dbg[dbg_n++] = ((long long)(slot) << 56) | (clock64() & 0x00ffffffffffffffll); \
    } while (0)

__device__ __forceinline__ const Lvl &find_level(const DcnArgs &a, int tile)
{
    int li = 0;
    while (li + 1 < a.nlv && tile >= a.lv[li + 1].tile0) ++li;
    return a.lv[li];
}

// Sample position exactly as the oracle / reference compute it:
//   py = float(ho*stride - pad + i*dil) * scale_h + dy     (kernel.cu:227-228, 281-282, 892-893)
// yx (optional): clamped corner rows / columns {cy0, cx0, cy1, cx1} of a sample with flags != 0
// The memory reads of a tap (offset pair, modulation scalar) and the arithmetic on them are separate steps so that a
// kernel can issue the loads a phase ahead of the table entry it builds from them.
struct TapRaw {
    float oy, ox, m;
};
__device__ __forceinline__ TapRaw tap_raw(const DcnArgs &a, const Lvl &L, int pix, int k, int dgi)
{
    TapRaw r = {0.f, 0.f, 1.f};   // L.off == NULL: a dense convolution (the regular grid)
    if (pix >= L.P) return r;
    const int K = a.kh * a.kw;
    const int HWo = L.Ho * L.Wo;
    const int b = pix / HWo;
    const int rem = pix - b * HWo;
    const int ho = rem / L.Wo, wo = rem - ho * L.Wo;
    if (L.off) {
        const float *op = L.off + (size_t)b * L.osb + (size_t)ho * L.osh + (size_t)wo * L.osw;
        r.oy = op[(size_t)(dgi * 2 * K + 2 * k) * L.osc];
        r.ox = op[(size_t)(dgi * 2 * K + 2 * k + 1) * L.osc];
    }
    if (L.msk) r.m = L.msk[(size_t)b * L.msb + (size_t)(dgi * K + k) * L.msc + (size_t)ho * L.msh + (size_t)wo * L.msw];
    return r;
}
__device__ __forceinline__ Tap tap_finish(const DcnArgs &a, const Lvl &L, int pix, int k, const TapRaw &raw, int4 *yx)
{
    Tap t;
    t.i00 = t.i01 = t.i10 = t.i11 = 0;
    t.ly = t.lx = t.m = 0.f;
    t.flags = 0;
    if (pix >= L.P) return t;
    const int HWo = L.Ho * L.Wo;
    const int b = pix / HWo;
    const int rem = pix - b * HWo;
    const int ho = rem / L.Wo, wo = rem - ho * L.Wo;
    const int i = k / a.kw, j = k - i * a.kw;
    float m = raw.m;
    if (a.msig && L.msk) m = 1.f / (1.f + expf(-m));
    const float py = __fadd_rn(__fmul_rn((float)(ho * a.stride - a.pad + i * a.dil), L.sh), raw.oy);
    const float px = __fadd_rn(__fmul_rn((float)(wo * a.stride - a.pad + j * a.dil), L.sw), raw.ox);
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
        if (yx) *yx = make_int4(cy0, cx0, cy1, cx1);
    }
    return t;
}
__device__ __forceinline__ Tap make_tap_ex(const DcnArgs &a, const Lvl &L, int pix, int k, int dgi, int4 *yx)
{
    return tap_finish(a, L, pix, k, tap_raw(a, L, pix, k, dgi), yx);
}

__device__ __forceinline__ Tap make_tap(const DcnArgs &a, const Lvl &L, int pix, int k, int dgi)
{
    return make_tap_ex(a, L, pix, k, dgi, nullptr);
}

__device__ __forceinline__ void corner_weights(const Tap &t, float &b00, float &b01, float &b10, float &b11)
{
    const float hy = 1.f - t.ly, hx = 1.f - t.lx;
    b00 = (t.flags & 1) ? hy * hx : 0.f;
    b01 = (t.flags & 2) ? hy * t.lx : 0.f;
    b10 = (t.flags & 4) ? t.ly * hx : 0.f;
    b11 = (t.flags & 8) ? t.ly * t.lx : 0.f;
}


// Sum over the 16 lanes of a DPP row (lanes 16q..16q+15) without touching the LDS: quad xor 1, quad xor 2,
// half-row mirror, row mirror.  (__shfl_xor compiles to ds_bpermute_b32 + a wait per step: 84 of them per chunk
// were the longest dependent chain of the backward-data epilogue.)
//
// Written as inline assembly with its own wait states between a step and whatever produced its input (the operands of
// an asm statement cannot be SLP-packed into v_pk_* registers; see the note on -fno-slp-vectorize in build.py).
#define LSN_DPP_ADD(nops, ctrl_text)                                                                       \
    asm("s_nop " nops "\n\tv_add_f32_dpp %0, %1, %1 " ctrl_text " row_mask:0xf bank_mask:0xf bound_ctrl:1" \
        : "=&v"(r)                                                                                        \
        : "v"(v))
__device__ __forceinline__ float row16_sum(float v)
{
    float r;
    LSN_DPP_ADD("7", "quad_perm:[1,0,3,2]");   // the input may come from a packed op: eight wait states
    v = r;
    LSN_DPP_ADD("4", "quad_perm:[2,3,0,1]");
    v = r;
    LSN_DPP_ADD("4", "row_half_mirror");
    v = r;
    LSN_DPP_ADD("4", "row_mirror");
    return r;
}
#undef LSN_DPP_ADD

// Branch-free guarded load of 4 consecutive floats p[0..3] of which the first `rem` (may be <= 0)
// are valid.  hipcc turns `if (cond) v = *ptr` into a branch with a full vmcnt(0) wait per load
// (cdna_hip_programming.md section 5, trap (c)), which serialises a staging phase into dependent L2
// round trips; here every lane always loads from a clamped, valid address and the result is
// selected afterwards.  VEC: the row is 16-byte aligned and rem is a multiple of 4.
template <bool VEC>
__device__ __forceinline__ float4 load4_guarded(const float *row, int off, int rem, bool row_ok)
{
    float4 v;
    if (VEC) {
        const bool ok = row_ok && rem > 0;
        v = *reinterpret_cast<const float4 *>(row + (ok ? off : 0));
        if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        const int base = (row_ok && rem > 0) ? off : 0;
        const int n = (row_ok && rem > 0) ? rem : 1;   // n >= 1 valid elements from base
        const float a = row[base];
        const float b = row[base + (n > 1 ? 1 : 0)];
        const float c = row[base + (n > 2 ? 2 : 0)];
        const float d = row[base + (n > 3 ? 3 : 0)];
        const bool ok = row_ok && rem > 0;
        v.x = ok ? a : 0.f;
        v.y = (ok && rem > 1) ? b : 0.f;
        v.z = (ok && rem > 2) ? c : 0.f;
        v.w = (ok && rem > 3) ? d : 0.f;
    }
    return v;
}

// chunk index -> (tap k, channel offset inside the conv group, #valid channels, deformable group)
struct Chunk {
    int k, c0, nval, dgi;
};
template <int CK>
__device__ __forceinline__ Chunk decode_chunk(const DcnArgs &a, int g, int t, int segs, int ncc)
{
    // order: tap-major, then segment, then CK-channel sub-chunk: all chunks of one (tap, deformable
    // group) are consecutive, so a thread keeps its sampling offsets / weights in registers across them
    // (measured: the channel-major order bought no L1 reuse)
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

// Walks the chunk sequence of decode_chunk without integer divisions (a division by a runtime value
// costs ~40 instructions; two decodes per chunk were ~20% of the pipelined kernel's chunk time).
// `next()` saturates at the last chunk.
template <int CK>
struct ChunkIter {
    int k, seg, cc, t, T, segs, ncc, SL, base_c, cpdg;
    __device__ __forceinline__ ChunkIter(const DcnArgs &a, int g, int segs_, int ncc_, int T_)
        : k(0), seg(0), cc(0), t(0), T(T_), segs(segs_), ncc(ncc_), SL(a.SL), base_c(g * (a.C / a.groups)),
          cpdg(a.C / a.dg) {}
    __device__ __forceinline__ Chunk get() const
    {
        // readfirstlane: the values are wave-uniform by construction; saying so keeps them in SGPRs (without
        // it hipcc wrapped the weight buffer loads in waterfall loops)
        Chunk c;
        c.k = __builtin_amdgcn_readfirstlane(k);
        c.c0 = __builtin_amdgcn_readfirstlane(seg * SL + cc * CK);
        c.nval = __builtin_amdgcn_readfirstlane(min(CK, SL - cc * CK));
        c.dgi = __builtin_amdgcn_readfirstlane((base_c + seg * SL) / cpdg);
        return c;
    }
    __device__ __forceinline__ void next()
    {
        if (t + 1 >= T) return;
        ++t;
        if (++cc == ncc) {
            cc = 0;
            if (++seg == segs) {
                seg = 0;
                ++k;
            }
        }
    }
};

// =============================================================================================
// Forward:  block = BM output pixels x BN output channels of one conv group; 4 waves (WM x WN),
// each owning TM x TN accumulator tiles of 32x32 (v_mfma_f32_32x32x2_f32).
//   LDS:  As[BM][33]  sampled+modulated values of the current chunk (pixel rows, k contiguous)
//         Bs[BN][33]  weight chunk (co rows, k contiguous);  stride 33 -> conflict-free both ways
//         tab[BM][K*dg] sampling table built once per block
// =============================================================================================
template <int BM, int BN, int WM, int WN, bool VEC>
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

    const int tile = xcd_remap(blockIdx.x, a.ntiles);
    const Lvl &L = find_level(a, tile);
    const int tile_p = (tile - L.tile0) * BM;
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
        const float *wbase = a.w + (size_t)co_base * Kdim + ch.k * Cg + ch.c0;
        const int rem = ch.nval - wq * 4;
#pragma unroll
        for (int ps = 0; ps < NPB; ++ps) {
            const int col = ps * 32 + wrow;
            const bool ok = col < nco;
            wv[ps] = load4_guarded<VEC>(wbase + (size_t)(ok ? col : 0) * Kdim, wq * 4, rem, ok);
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

    int dbg_n = 0;
    LSN_STAMP(0);
    __syncthreads();  // sampling table complete
    LSN_STAMP(1);
    load_chunk(0);
    for (int t = 0; t < T; ++t) {
        LSN_STAMP(2);
        // staging (VALU / LDS / VMEM issue) gets issue priority over the co-resident block's MFMA
        // phase: measured on MI355X, a wave in a back-to-back MFMA stream otherwise starves its SIMD
        // partner's staging (store phase 2.3k -> 5.7k cycles), serialising the two blocks of a CU
        __builtin_amdgcn_s_setprio(2);
        store_chunk(t);
        LSN_STAMP(3);
        __syncthreads();
        LSN_STAMP(4);
        if (t + 1 < T) load_chunk(t + 1);  // in flight during the MFMA phase below
        __builtin_amdgcn_s_setprio(0);
        LSN_STAMP(5);
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
        LSN_STAMP(6);
        __syncthreads();
        LSN_STAMP(7);
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

// Raw buffer loads (32-bit per-lane offset + scalar offset, the hardware bounds check returns 0 beyond num_records) and the
// tile constants of the one-workgroup-per-CU kernels.  (Rounds 1 - 5 kept an fp32 software-pipelined forward here,
// dcn_fwd_pipe_kernel; round 6 removed it with the windowed scatter kernels: exact fp32 is dcn_fwd_kernel.)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float buf_load_f32(__amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}
__device__ __forceinline__ float4 buf_load_f32x4(__amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    // NB: assigning the builtin's result to a 4 x u32 ext-vector and indexing it makes hipcc (ROCm 7.2)
    // emit buffer_load_dword + splat; copying the 16 bytes out keeps the dwordx4 load.
    auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    static_assert(sizeof(v) == 16, "b128");
    float4 f;
    __builtin_memcpy(&f, &v, 16);
    return f;
}

__device__ __forceinline__ float2 buf_load_f32x2(__amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    auto v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
    static_assert(sizeof(v) == 8, "b64");
    float2 f;
    __builtin_memcpy(&f, &v, 8);
    return f;
}

constexpr int PIPE_BM = 64, PIPE_BN = 256, PIPE_BK = 32, PIPE_LDK = 36;

// =============================================================================================
// Forward on the bf16 matrix pipe with split operands (dcn_fwd_xn_kernel<PREP, NP>).
//
// Same tiling and software pipeline as dcn_fwd_pipe_kernel (64 px x 256 co, one workgroup per CU, chunk t's
// MFMAs interleaved with the LDS commit of chunk t+1 and the load issue of chunk t+2), but the blended samples and
// the weights are split into bf16 planes while they are staged (common.h: NP = 3 products on 2 planes, or NP = 6
// products on 3 planes = fp32-equivalent) and multiplied on v_mfma_f32_32x32x16_bf16.  4 NP MFMAs of 32 cycles per
// k-step instead of 64 of 64 cycles per chunk, and the VALU work of the staging overlaps them (the bf16 pipe is
// separate from the fp32 ALUs).
// LDS rows: 32 k-values = 64 B of bf16, no padding; the four 16-byte slots of row r are stored at slot ^ ((r >> 2) & 3),
// which makes the 16-byte operand reads of every ds_read_b128 lane group cover all 64 banks exactly once (rows with
// equal r & 3 inside a group differ in (r >> 2) & 3).  One plane per array: buffer b = [A planes][B planes].
// =============================================================================================
constexpr int XN_RS = 64;   // LDS row stride, bytes

template <int NP>
__host__ __device__ inline size_t xn_lds_bytes(int KD)
{
    return (size_t)2 * SplitCfg<NP>::NPL * (PIPE_BM + PIPE_BN) * XN_RS + (size_t)PIPE_BM * KD * sizeof(Tap);
}

// byte offset of 16-byte slot `slot` of row `row` inside a plane
__device__ __forceinline__ int xn_slot(int row, int slot) { return row * XN_RS + ((slot ^ ((row >> 2) & 3)) << 4); }

// PREP: a.wtp holds the weights already split by dcn_prepare_w_kernel (NPL planes of [Co][K][C/groups] bf16):
// their staging is a 16-byte copy per slice instead of ~14 VALU instructions per float4
template <bool PREP, int NP>
__global__ __launch_bounds__(256, 1) void dcn_fwd_xn_kernel(const DcnArgs a)
{
    using SC = SplitCfg<NP>;
    constexpr int NPL = SC::NPL;
    constexpr int BM = PIPE_BM, BN = PIPE_BN, BK = PIPE_BK, RS = XN_RS;
    constexpr int NPA = BM / 16;                       // 4 gather passes (2 channels per thread)
    constexpr int NPB = PREP ? NPL * (BN / 64) : BN / 32;   // weight slices of 256 threads x 16 B
    constexpr int PLANE_A = BM * RS, PLANE_B = BN * RS, BUF = NPL * (PLANE_A + PLANE_B);
    // micro-slots of a chunk's staging: 5 per pixel slice (blend, split A, split B, LDS writes, loads of chunk t+2),
    // 2 per weight slice; spread over the NGAP gaps between the MFMAs (see staging_slot)
    constexpr int NGAP = 2 * NP * 4, NSLOT = 5 * NPA + 2 * NPB;
    extern __shared__ __align__(16) unsigned char smem[];
    Tap *tab = reinterpret_cast<Tap *>(smem + 2 * BUF);   // [BM][K*dg]

    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int K = a.kh * a.kw, KD = K * a.dg;
    const int Cg = a.C / a.groups, Cog = a.Co / a.groups, Kdim = K * Cg;

    const int tile = xcd_remap(blockIdx.x, a.ntiles);
    const Lvl &L = find_level(a, tile);
    const int tile_p = (tile - L.tile0) * BM;
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
    const int kk2 = tid & 15, prow = tid >> 4;   // gather: channel pair, pixel row (16 rows per pass)
    const int wq = tid & 7, wrow = tid >> 3;     // weights: float4 slot along k, co row (32 rows per pass)

    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(L.x), 0, L.B * L.H * L.W * a.C * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs =
        PREP ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(a.wtp), 0, a.Co * Kdim * 2 * NPL, 0x00020000)
             : __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.w), 0, a.Co * Kdim * 4, 0x00020000);
    // PREP slices: plane ps % NPL, rows (ps / NPL) * 64 + (tid >> 2), 16-byte slot tid & 3
    const int pq = tid & 3, prw = tid >> 2;

    int wvoff[NPB];
#pragma unroll
    for (int ps = 0; ps < NPB; ++ps) {
        if (PREP) {
            const int col = (ps / NPL) * 64 + prw;
            wvoff[ps] = (col < nco) ? ((ps % NPL) * a.Co * Kdim + (co_base + col) * Kdim) * 2 + pq * 16 : 0x7ffffff0;
        } else {
            const int col = ps * 32 + wrow;
            wvoff[ps] = (col < nco) ? ((co_base + col) * Kdim + wq * 4) * 4 : 0x7ffffff0;
        }
    }
    // LDS commit addresses of this thread (the row swizzle depends on the thread's row only)
    const int xcommit = xn_slot(prow, kk2 >> 2) + (kk2 & 3) * 4;                    // + ps * 16 * RS
    const int wcommit = PREP ? xn_slot(prw, pq) : xn_slot(wrow, wq >> 1) + (wq & 1) * 8;   // + row block

    int voffI[NPA][4];
    float wgtC[NPA][4];
    auto load_offsets = [&](const Chunk &ch) {
#pragma unroll
        for (int ps = 0; ps < NPA; ++ps) {
            const int4 idx = *reinterpret_cast<const int4 *>(&tab[(ps * 16 + prow) * KD + ch.dgi * K + ch.k]);
            voffI[ps][0] = (idx.x + 2 * kk2) * 4;
            voffI[ps][1] = (idx.y + 2 * kk2) * 4;
            voffI[ps][2] = (idx.z + 2 * kk2) * 4;
            voffI[ps][3] = (idx.w + 2 * kk2) * 4;
        }
    };
    auto load_weights = [&](const Chunk &ch) {
#pragma unroll
        for (int ps = 0; ps < NPA; ++ps) {
            const Tap tp = tab[(ps * 16 + prow) * KD + ch.dgi * K + ch.k];
            float b00, b01, b10, b11;
            corner_weights(tp, b00, b01, b10, b11);
            wgtC[ps][0] = b00 * tp.m;
            wgtC[ps][1] = b01 * tp.m;
            wgtC[ps][2] = b10 * tp.m;
            wgtC[ps][3] = b11 * tp.m;
        }
    };

    float2 xv[NPA][4];
    float4 wv[NPB];
    auto issue_x = [&](const Chunk &ch, int ps) {
        const int soff = (g * Cg + ch.c0) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) xv[ps][q] = buf_load_f32x2(xrs, voffI[ps][q], soff);
    };
    auto issue_w = [&](const Chunk &ch, int ps) {
        wv[ps] = buf_load_f32x4(wrs, wvoff[ps], (ch.k * Cg + ch.c0) * (PREP ? 2 : 4));
    };
    auto commit_x = [&](const Chunk &ch, int ps, unsigned char *buf) {
        float v0 = wgtC[ps][0] * xv[ps][0].x + wgtC[ps][1] * xv[ps][1].x + wgtC[ps][2] * xv[ps][2].x +
                   wgtC[ps][3] * xv[ps][3].x;
        float v1 = wgtC[ps][0] * xv[ps][0].y + wgtC[ps][1] * xv[ps][1].y + wgtC[ps][2] * xv[ps][2].y +
                   wgtC[ps][3] * xv[ps][3].y;
        v0 = (2 * kk2 < ch.nval) ? v0 : 0.f;
        v1 = (2 * kk2 + 1 < ch.nval) ? v1 : 0.f;
        unsigned pl[NPL];
        split_planes<NPL>(v0, v1, pl);
        unsigned char *p = buf + ps * 16 * RS + xcommit;
#pragma unroll
        for (int q = 0; q < NPL; ++q) *reinterpret_cast<unsigned *>(p + q * PLANE_A) = pl[q];
    };
    auto commit_w = [&](const Chunk &ch, int ps, unsigned char *buf) {
        if (PREP) {   // columns past nval hold neighbouring values; the A operand is zero there
            unsigned char *p = buf + NPL * PLANE_A + (ps % NPL) * PLANE_B + (ps / NPL) * 64 * RS + wcommit;
            *reinterpret_cast<float4 *>(p) = wv[ps];
            return;
        }
        const bool ok = wq * 4 < ch.nval;   // nval is a multiple of 4 on this path (vec_ok)
        const float4 v = ok ? wv[ps] : make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned p0[NPL], p1[NPL];
        split_planes<NPL>(v.x, v.y, p0);
        split_planes<NPL>(v.z, v.w, p1);
        unsigned char *p = buf + NPL * PLANE_A + ps * 32 * RS + wcommit;
#pragma unroll
        for (int q = 0; q < NPL; ++q) *reinterpret_cast<uint2 *>(p + q * PLANE_B) = make_uint2(p0[q], p1[q]);
    };
    // Staging slot s of a chunk.  The wave owns its SIMD alone, so what does not fit into the 32-cycle shadow of an
    // MFMA stalls the matrix pipe (dense-conv kernel, tools/phase_clocks.py conv: 48 MFMAs alone 1.64 k cycles, 2.8 k
    // with whole slices in the gaps, 1.9 k with micro-slots).  A pixel slice is therefore cut into five slots of a few
    // instructions -- bilinear blend, split step A (hi plane + residual), split step B (mid / lo planes), the LDS
    // writes, the four corner loads of chunk t+2 into the registers just consumed -- then the weight slices
    // (commit, issue).
    float sp_v0 = 0.f, sp_v1 = 0.f, sp_r0 = 0.f, sp_r1 = 0.f;
    unsigned sp_h = 0, sp_m = 0, sp_l = 0;
    auto staging_slot = [&](int s, const Chunk &c1, const Chunk &c2, unsigned char *bn) {
        if (s < 5 * NPA) {
            const int ps = s / 5, part = s - 5 * ps;
            if (part == 0) {
                const float v0 = wgtC[ps][0] * xv[ps][0].x + wgtC[ps][1] * xv[ps][1].x + wgtC[ps][2] * xv[ps][2].x +
                                 wgtC[ps][3] * xv[ps][3].x;
                const float v1 = wgtC[ps][0] * xv[ps][0].y + wgtC[ps][1] * xv[ps][1].y + wgtC[ps][2] * xv[ps][2].y +
                                 wgtC[ps][3] * xv[ps][3].y;
                sp_v0 = (2 * kk2 < c1.nval) ? v0 : 0.f;
                sp_v1 = (2 * kk2 + 1 < c1.nval) ? v1 : 0.f;
            } else if (part == 1) {
                const bf16x2 h = {(__bf16)sp_v0, (__bf16)sp_v1};
                sp_h = __builtin_bit_cast(unsigned, h);
                sp_r0 = sp_v0 - __uint_as_float(sp_h << 16);
                sp_r1 = sp_v1 - __uint_as_float(sp_h & 0xffff0000u);
            } else if (part == 2) {
                const bf16x2 m = {(__bf16)sp_r0, (__bf16)sp_r1};
                sp_m = __builtin_bit_cast(unsigned, m);
                if constexpr (NPL == 3) {
                    const float s0 = sp_r0 - __uint_as_float(sp_m << 16), s1 = sp_r1 - __uint_as_float(sp_m & 0xffff0000u);
                    const bf16x2 l = {(__bf16)s0, (__bf16)s1};
                    sp_l = __builtin_bit_cast(unsigned, l);
                }
            } else if (part == 3) {
                unsigned char *p = bn + ps * 16 * RS + xcommit;
                *reinterpret_cast<unsigned *>(p) = sp_h;
                *reinterpret_cast<unsigned *>(p + PLANE_A) = sp_m;
                if constexpr (NPL == 3) *reinterpret_cast<unsigned *>(p + 2 * PLANE_A) = sp_l;
            } else {
                issue_x(c2, ps);
            }
        } else {
            const int sw = s - 5 * NPA, ps = sw >> 1;
            if ((sw & 1) == 0)
                commit_w(c1, ps, bn);
            else
                issue_w(c2, ps);
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

    __syncthreads();  // sampling table complete
    ChunkIter<BK> it1(a, g, segs, ncc, T), it2(a, g, segs, ncc, T);
    Chunk cI = it1.get();
    Chunk cC = cI;
    load_offsets(cI);
    load_weights(cC);
    {
#pragma unroll
        for (int ps = 0; ps < NPA; ++ps) issue_x(cI, ps);
#pragma unroll
        for (int ps = 0; ps < NPB; ++ps) issue_w(cI, ps);
#pragma unroll
        for (int ps = 0; ps < NPA; ++ps) commit_x(cC, ps, smem);
#pragma unroll
        for (int ps = 0; ps < NPB; ++ps) commit_w(cC, ps, smem);
        it1.next();
        it2.next();
        it2.next();
        const Chunk c1 = it1.get();
        if (c1.k != cI.k || c1.dgi != cI.dgi) load_offsets(c1);
        cI = c1;
#pragma unroll
        for (int ps = 0; ps < NPA; ++ps) issue_x(c1, ps);
#pragma unroll
        for (int ps = 0; ps < NPB; ++ps) issue_w(c1, ps);
    }
    __syncthreads();

    // operand read offsets: row (lane & 31) of a 32-row block, k-half (lane >> 5), k-step ks: slot (lane >> 5) + 2 ks
    const int rsw = (lane >> 2) & 3;
    const int ro0 = (lane & 31) * RS + ((((lane >> 5)) ^ rsw) << 4), ro1 = ro0 ^ 32;

    int dbg_n = 0;
    for (int t = 0; t < T; ++t) {
        LSN_STAMP(2);
        const int cur = t & 1;
        const unsigned char *bc = smem + cur * BUF;
        unsigned char *bn = smem + (cur ^ 1) * BUF;
        const Chunk c1 = it1.get();
        const Chunk c2 = it2.get();
        it1.next();
        it2.next();
        if (c1.k != cC.k || c1.dgi != cC.dgi) load_weights(c1);
        cC = c1;
        if (c2.k != cI.k || c2.dgi != cI.dgi) load_offsets(c2);
        cI = c2;

        // operands of both k-steps: [ks][tile row/col block][plane]
        const unsigned char *ap = bc;
        const unsigned char *bp = bc + NPL * PLANE_A + wn * 64 * RS;
        bf16x8 Af[2][2][NPL], Bf[2][2][NPL];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const int ro = ks ? ro1 : ro0;
                    Af[ks][i][q] = *reinterpret_cast<const bf16x8 *>(ap + q * PLANE_A + i * 32 * RS + ro);
                    Bf[ks][i][q] = *reinterpret_cast<const bf16x8 *>(bp + q * PLANE_B + i * 32 * RS + ro);
                }
        if (a.dbg != nullptr) {
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): diagnostic only
            LSN_STAMP(5);
        }
        // NGAP MFMAs; the staging slots of chunk t+1 / t+2 are spread over the gaps between them
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
#pragma unroll
                    for (int s = gap * NSLOT / NGAP; s < (gap + 1) * NSLOT / NGAP; ++s) staging_slot(s, c1, c2, bn);
                    __builtin_amdgcn_sched_barrier(0);
                }
        LSN_STAMP(6);
        __syncthreads();
        LSN_STAMP(7);
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = wn * 64 + j * 32 + (lane & 31);
            if (col >= nco) continue;
            const float bv = a.bias ? a.bias[co_base + col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pix = tile_p + i * 32 + mfma32_row(r, lane);
                if (pix < L.P) L.out[(size_t)pix * a.Co + co_base + col] = (acc[i][j][r] + accl[i][j][r]) + bv;
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
template <int RED, bool VEC>
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

    const int tile = xcd_remap(blockIdx.x, a.ntiles);
    const Lvl &L = find_level(a, tile);
    const int tile_p = (tile - L.tile0) * BWD_BM;

    for (int e = tid; e < BWD_BM * KD; e += 256) {
        const int pl = e / KD, r = e - pl * KD;
        const int dgi = r / K, k = r - dgi * K;
        tab[e] = make_tap(a, L, tile_p + pl, k, dgi);
    }
    for (int e = tid; e < BWD_BM * KD * 3; e += 256) gacc[e] = 0.f;

    const int segs = Cg / a.SL, ncc = (a.SL + BK - 1) / BK;
    const int T = K * segs * ncc;
    const int nrb = (Cog + RED - 1) / RED;  // reduction blocks (1 for Co/groups <= RED)
    // bits 26..28 of the debug word: timing ablations (tools/phase_clocks.py bwd); results are wrong when set
    const bool abl_no_atomic = (a.dbg_block >> 26) & 1;
    const bool want_off = ((L.goff != nullptr) || (L.gmsk != nullptr)) && !((a.dbg_block >> 27) & 1);

    const int wq = tid & 7, wrow = tid >> 3;  // weight staging: float4 slot along k, 32 rows/pass
    constexpr int NPB = RED / 32;

    const int my_pix = tile_p + wave * 16 + j16;  // A operand row (pixel) of this lane

    float areg[QR];
    float4 wv[NPB];
    __syncthreads();

    // A operand: gout[pix][g*Cog + rb*RED + kq*QR + s], s = 0..QR-1 (this lane's k-quarter)
    auto load_a = [&](int g, int rb) {
        const int cb = rb * RED + kq * QR;
        const bool pix_ok = my_pix < L.P;
        const float *grow = L.gout + (size_t)(pix_ok ? my_pix : 0) * a.Co + g * Cog;
#pragma unroll
        for (int s4 = 0; s4 < QR / 4; ++s4) {
            const float4 v = load4_guarded<VEC>(grow, cb + s4 * 4, Cog - (cb + s4 * 4), pix_ok);
            areg[s4 * 4 + 0] = v.x;
            areg[s4 * 4 + 1] = v.y;
            areg[s4 * 4 + 2] = v.z;
            areg[s4 * 4 + 3] = v.w;
        }
    };
    // weight slab of (chunk t, reduction block rb): rows = output channels, 32 k-columns
    auto load_w = [&](int g, int t, int rb) {
        const Chunk ch = decode_chunk<BK>(a, g, t, segs, ncc);
        const float *wbase = a.w + (size_t)(g * Cog + rb * RED) * Kdim + ch.k * Cg + ch.c0;
        const int rem = ch.nval - wq * 4;
#pragma unroll
        for (int ps = 0; ps < NPB; ++ps) {
            const int row = ps * 32 + wrow;
            const bool ok = rb * RED + row < Cog;
            wv[ps] = load4_guarded<VEC>(wbase + (size_t)(ok ? row : 0) * Kdim, wq * 4, rem, ok);
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

    int dbg_n = 0;
    for (int g = 0; g < a.groups; ++g) {
        if (nrb == 1) load_a(g, 0);
        load_w(g, 0, 0);
        // offset / mask gradient partial sums of this lane's 4 pixel rows, carried over the chunks of one
        // (tap, deformable group) and reduced across the 16 channel lanes once, when the tap is finished
        float sy[4] = {0.f, 0.f, 0.f, 0.f}, sx[4] = {0.f, 0.f, 0.f, 0.f}, sm[4] = {0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < T; ++t) {
            const Chunk ch = decode_chunk<BK>(a, g, t, segs, ncc);
            const Chunk chn = decode_chunk<BK>(a, g, min(t + 1, T - 1), segs, ncc);
            const bool tap_done = (t + 1 == T) || chn.k != ch.k || chn.dgi != ch.dgi;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            for (int rb = 0; rb < nrb; ++rb) {
                if (nrb > 1) load_a(g, rb);  // Co/groups > RED: re-read the gout rows per slab
                LSN_STAMP(2);
                store_w();
                LSN_STAMP(3);
                __syncthreads();
                LSN_STAMP(4);
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
                LSN_STAMP(5);
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
                    if (L.gx != nullptr && cval && tp.flags && !abl_no_atomic) {
                        // one guard per sample, not per corner: a clamped corner of a border sample gets +0 at
                        // a valid neighbour address (wholly invalid samples must be skipped: their index is 0
                        // and thousands of same-address atomics serialise in L2)
                        atomic_add_f32(L.gx + tp.i00 + c, b00 * gm);
                        atomic_add_f32(L.gx + tp.i01 + c, b01 * gm);
                        atomic_add_f32(L.gx + tp.i10 + c, b10 * gm);
                        atomic_add_f32(L.gx + tp.i11 + c, b11 * gm);
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
                        sy[r] += gm * dy;
                        sx[r] += gm * dx;
                        sm[r] += gval * bil;
                    }
                }
            }
            if (want_off && tap_done) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float vy = row16_sum(sy[r]), vx = row16_sum(sx[r]), vm = row16_sum(sm[r]);
                    if (j16 == 0) {   // the row is private to this wave; conv groups sharing a deformable
                        float *ga = gacc + ((wave * 16 + kq * 4 + r) * KD + ch.dgi * K + ch.k) * 3;   // group add up
                        ga[0] += vy;
                        ga[1] += vx;
                        ga[2] += vm;
                    }
                    sy[r] = sx[r] = sm[r] = 0.f;
                }
            }
            LSN_STAMP(6);
            __syncthreads();  // Bs free for the next chunk
            LSN_STAMP(7);
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
// Backward-data on the bf16 matrix pipe (dcn_bwd_data_xn_kernel<NP, COLBUF>), groups = 1.
//
// The column-gradient GEMM of dcn_bwd_data_kernel as split-bf16 products on v_mfma_f32_16x16x32_bf16: the gout rows are
// split once per tile into bf16 plane registers, the weight slab of a chunk comes from planes that
// dcn_prepare_wt_kernel has split AND transposed to [tap][ci][co] (co contiguous = the MFMA k index) and is DMA'd
// into LDS (buffer_load ... lds; rows unpadded, 16-byte slots XOR-swizzled on the source side).
// COLBUF = true (default, see "grad_input without atomics" below): the mask-weighted column gradient is stored once
// per element into a.gcol and grad_input is produced by dcn_gather_kernel; the epilogue keeps only the four corner
// sums H = sum_c colgrad[c] x[corner, c] for the offset / mask gradients.
// COLBUF = false: the merged atomic scatter of round 1 -- the 4 pixels a lane owns are x-adjacent, equal corner
// addresses are summed in registers before one fp32 atomic per run (exact-mode / A/B runs: LSNET_BWD_GATHER=0).
// grid.y = tap groups: see the kernel.
constexpr int BX3_RS = 528;   // bytes per LDS row of the transposed weight slab: 256 co x 2 B + 16 B pad



// w (n floats, any layout) -> NPL bf16 planes of n values each (hi, [mid,] lo), same element order
template <int NPL>
__global__ void dcn_prepare_w_kernel(const float *w, unsigned short *out, size_t n)
{
    for (size_t e = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * 2; e < n; e += (size_t)gridDim.x * blockDim.x * 2) {
        unsigned pl[NPL];
        split_planes<NPL>(w[e], w[e + 1], pl);
#pragma unroll
        for (int q = 0; q < NPL; ++q) *reinterpret_cast<unsigned *>(out + q * n + e) = pl[q];
    }
}

// w (Co, K, C) fp32 -> NPL planes [K][C][Co] bf16
template <int NPL>
__global__ void dcn_prepare_wt_kernel(const float *w, unsigned short *out, int Co, int K, int C)
{
    const size_t n = (size_t)Co * K * C;
    for (size_t e = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * 2; e < n; e += (size_t)gridDim.x * blockDim.x * 2) {
        // destination index e = (k * C + ci) * Co + co, two consecutive co per thread
        const int co = (int)(e % Co);
        const size_t r = e / Co;
        const int ci = (int)(r % C), k = (int)(r / C);
        const float v0 = w[((size_t)co * K + k) * C + ci], v1 = w[((size_t)(co + 1) * K + k) * C + ci];
        unsigned pl[NPL];
        split_planes<NPL>(v0, v1, pl);
#pragma unroll
        for (int q = 0; q < NPL; ++q) *reinterpret_cast<unsigned *>(out + q * n + e) = pl[q];
    }
}

// NP: bf16 products per fp32 product (common.h).  COLBUF: instead of scattering into grad_input with atomics the
// mask-weighted column gradients go to a.gcol (one 64-byte run per pixel and half-slab), and grad_input is formed by
// dcn_gather_kernel from per-anchor sample lists: no atomics, each grad_input element written once, fixed summation
// order.  grad_offset / grad_mask are produced here in both variants.
//
// Weight slab [NPL planes][32 channel rows][256 co] bf16: loaded global -> LDS directly (buffer_load ... lds, no staging
// registers, no ds_write pass).  Rows are 512 B without padding; 16-byte slot u of row r lives at slot u ^ (r & 15)
// (applied on the SOURCE address of the load and on the read), which spreads the 16 rows x 4 k-quarters of every
// ds_read_b128 lane group over all 64 banks.  Per chunk: wait for the slab, MFMAs, barrier, issue the next slab's
// loads, epilogue (the loads land meanwhile).
// Accumulation: the leading product h*h and the five small products go to separate accumulators that are added once
// per chunk, so the fp32 rounding of the running sum is not paid once per small term.
constexpr int BXN_ROW = 512;   // bytes per LDS row of the transposed weight slab (256 co x 2 B)

__host__ __device__ inline size_t bwd_xn_lds_bytes(int np, int KD)
{
    return (size_t)(np == 6 ? 3 : 2) * 32 * BXN_ROW + (size_t)BWD_BM * KD * (sizeof(Tap) + 12);
}

template <int NP, bool COLBUF>
__global__ __launch_bounds__(256, 2) void dcn_bwd_data_xn_kernel(const DcnArgs a)
{
    using SC = SplitCfg<NP>;
    constexpr int NPL = SC::NPL;
    constexpr int BK = 32, RED = 256, NS = RED / 32, ROW = BXN_ROW, PLANE = 32 * ROW;
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char *Bp = smem;                                            // [NPL][32 ch][RED co] bf16 planes, swizzled
    Tap *tab = reinterpret_cast<Tap *>(smem + NPL * PLANE);              // [64][K*dg]
    const int K = a.kh * a.kw, KD = K * a.dg;
    float *gacc = reinterpret_cast<float *>(tab + BWD_BM * KD);          // [64][KD][3]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j16 = lane & 15, kq = lane >> 4;
    const int C = a.C, Co = a.Co;   // groups == 1

    // (tile, tap group) from the XCD-ordered linear workgroup id: the tap groups of a tile and the neighbouring tiles
    // stay behind one L2 (they read the same grad_output rows and the same input rows)
    const int TG = (int)gridDim.y;
    const int work = xcd_remap((int)(blockIdx.y * gridDim.x + blockIdx.x), a.ntiles * TG);
    const int tile = work / TG, tgi = work - tile * TG;
    const Lvl &L = find_level(a, tile);
    const int tile_p = (tile - L.tile0) * BWD_BM;

    if (COLBUF && a.gtap != nullptr) {   // the launch-wide table of dcn_bin_kernel (k-major): per tap, 64 entries in a row
        const uint4 *src = reinterpret_cast<const uint4 *>(a.gtap + (size_t)(L.prow0 + tile_p));   // 16-byte halves
        const int nvalid = min(BWD_BM, L.P - tile_p);   // tile rows past the level end stay zero = no sample
        for (int e = tid; e < BWD_BM * KD * 2; e += 256) {
            const int ent = e >> 1, kd = ent / BWD_BM, pl = ent - kd * BWD_BM;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (pl < nvalid) v = src[((size_t)kd * a.gtap_rows + pl) * 2 + (e & 1)];
            reinterpret_cast<uint4 *>(tab)[(pl * KD + kd) * 2 + (e & 1)] = v;
        }
    } else {
        for (int e = tid; e < BWD_BM * KD; e += 256) {
            const int pl = e / KD, r = e - pl * KD;
            const int dgi = r / K, k = r - dgi * K;
            tab[e] = make_tap(a, L, tile_p + pl, k, dgi);
        }
    }
    for (int e = tid; e < BWD_BM * KD * 3; e += 256) gacc[e] = 0.f;

    const int segs = C / a.SL, ncc = (a.SL + BK - 1) / BK;
    // tap group tgi of TG (= gridDim.y): this block's share of the TAPS (all chunks of taps [k_lo, k_hi)).  Every chunk writes its
    // own slice of the column gradients and a tap's offset / mask gradients are complete inside one block, so the split
    // needs no atomics; it exists to make the blocks short enough to fill the tail of the launch (699 tiles on 512
    // resident blocks ran as two rounds, the second 37 % full).
    const int k_lo = K * tgi / TG, k_hi = K * (tgi + 1) / TG;
    const int T = (k_hi - k_lo) * segs * ncc;
    const bool want_off = (L.goff != nullptr) || (L.gmsk != nullptr);
    const bool want_gx = COLBUF ? (a.gcol != nullptr && L.gx != nullptr) : (L.gx != nullptr);

    // A operand: gout row of pixel (wave * 16 + j16), k-step s covers co = 32 s + 8 kq .. + 7, split once
    bf16x8 af[NS][NPL];
    {
        const int my_pix = tile_p + wave * 16 + j16;
        const bool pix_ok = my_pix < L.P;
        const float *grow = L.gout + (size_t)(pix_ok ? my_pix : 0) * Co;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int cb = s * 32 + kq * 8;
            float v[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bool ok = pix_ok && cb + h * 4 < Co;   // Co % 4 == 0 on this path
                const float4 f = *reinterpret_cast<const float4 *>(grow + (ok ? cb + h * 4 : 0));
                v[h * 4 + 0] = ok ? f.x : 0.f, v[h * 4 + 1] = ok ? f.y : 0.f, v[h * 4 + 2] = ok ? f.z : 0.f,
                          v[h * 4 + 3] = ok ? f.w : 0.f;
            }
            unsigned pl[4][NPL];
#pragma unroll
            for (int e = 0; e < 4; ++e) split_planes<NPL>(v[2 * e], v[2 * e + 1], pl[e]);
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                const uint4 U = make_uint4(pl[0][q], pl[1][q], pl[2][q], pl[3][q]);
                __builtin_memcpy(&af[s][q], &U, 16);
            }
        }
    }

    // weight slab: pass ps = (plane, 8-row group); thread = (row tid >> 5 of the group, 16-byte LDS slot tid & 31).
    // The LDS destination of a wave is its 1 KB piece of the pass (lane-linear); LDS slot t of row r receives the
    // source slot t ^ (r & 15).  Rows past nval and co slots past Co read a valid neighbour instead (their products
    // meet a zero A operand resp. an ignored output column), so every LDS byte is (re)written with finite data.
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(a.wtp), 0,
                                                                         K * C * Co * 2 * NPL, 0x00020000);
    const int co_slots = Co >> 3;   // Co % 8 == 0 on this path
    auto issue_w = [&](const Chunk &ch) {
        const int rowbase = ch.k * C + ch.c0;   // row index into [K*C][Co]
#pragma unroll
        for (int ps = 0; ps < 4 * NPL; ++ps) {
            const int plane = ps >> 2, r = (ps & 3) * 8 + (tid >> 5);
            const int u = (tid & 31) ^ (r & 15);
            const int voff = (plane * K * C * Co + (rowbase + min(r, ch.nval - 1)) * Co) * 2 + min(u, co_slots - 1) * 16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                wrs, (__attribute__((address_space(3))) void *)(Bp + plane * PLANE + (ps & 3) * 4096 + wave * 1024), 16,
                voff, 0, 0, 0);
        }
    };

    __syncthreads();
    int dbg_n = 0;
    ChunkIter<BK> it(a, 0, segs, ncc, T);
    it.k = k_lo;   // (t counts from 0 inside the block's range)
    Chunk ch = it.get();
    issue_w(ch);
    // operand reads: row j16 (channel cl = j16) and row 16 + j16, source slot 4 s + kq -> LDS slot (4 s + kq) ^ j16
    const unsigned char *b0 = Bp + j16 * ROW, *b1 = Bp + (16 + j16) * ROW;
    const int trow = (wave * 16 + kq * 4) * KD;   // tap-table row of this lane's first pixel
    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(L.x), 0, L.B * L.H * L.W * C * 4, 0x00020000);
    // column gradients of this level: rows (pix * K + k) of C floats; rows of pixels past the level end fall outside
    // num_records and are dropped by the hardware
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(
        COLBUF && a.gcol ? a.gcol + (size_t)L.prow0 * K * C : const_cast<float *>(L.x), 0,
        COLBUF && want_gx ? L.P * K * C * 4 : 0, 0x00020000);
    const int grow0 = (tile_p + wave * 16 + kq * 4) * K * C * 4;   // byte offset of this lane's first pixel, tap 0

    // grad_offset / grad_mask of a sample are bilinear forms of the four dot products
    //     H_corner = sum over channels of gcol[c] * x[corner, c]
    // so only these are accumulated per channel (4 fmas; carried over the slabs of a tap in registers); the corner
    // validity, the fractions and the modulation scalar are applied once per (pixel, tap) after the 16 channel lanes
    // have been summed.  Invalid corners are loaded from their clamped (valid) addresses and get weight 0 there.
    // The corner values are loaded a phase ahead of their use: half-slab 0 before the MFMA block, half-slab 1 before
    // half-slab 0 is consumed (a dependent L2 round trip per half-slab was most of the epilogue).
    float H[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) H[r][0] = H[r][1] = H[r][2] = H[r][3] = 0.f;
    auto issue_xv = [&](const Chunk &c_, int tn, float (&xv)[4][4]) {
        const int cl = tn * 16 + j16;
        const int cb = (c_.c0 + (cl < c_.nval ? cl : 0)) * 4;
        const int kd = c_.dgi * K + c_.k;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int4 idx = *reinterpret_cast<const int4 *>(&tab[trow + r * KD + kd]);
            xv[r][0] = buf_load_f32(xrs, idx.x * 4 + cb, 0);
            xv[r][1] = buf_load_f32(xrs, idx.y * 4 + cb, 0);
            xv[r][2] = buf_load_f32(xrs, idx.z * 4 + cb, 0);
            xv[r][3] = buf_load_f32(xrs, idx.w * 4 + cb, 0);
        }
    };
    // consume gcol[16 px][16 ch] of half-slab tn: D row = 4*kq + r (x-adjacent pixels), col = tn*16 + j16
    auto consume = [&](const Chunk &c_, int tn, const f32x4 &acc, const float (&xv)[4][4]) {
        const int kd = c_.dgi * K + c_.k;
        const int cl = tn * 16 + j16;
        const bool cval = cl < c_.nval;
        const int c = c_.c0 + (cval ? cl : 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float gval = cval ? acc[r] : 0.f;
            if (want_off) {
                H[r][0] += gval * xv[r][0];
                H[r][1] += gval * xv[r][1];
                H[r][2] += gval * xv[r][2];
                H[r][3] += gval * xv[r][3];
            }
            if (COLBUF) {
                if (want_gx && cval) {
                    const float m = tab[trow + r * KD + kd].m;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gval * m), grs,
                                                          grow0 + (r * K + c_.k) * C * 4 + c * 4, 0, 0);
                }
            }
        }
        if (!COLBUF && want_gx && cval) {
            Tap tp[4];
            float gm[4], w00[4], w01[4], w10[4], w11[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                tp[r] = tab[trow + r * KD + kd];
                corner_weights(tp[r], w00[r], w01[r], w10[r], w11[r]);
                gm[r] = acc[r] * tp[r].m;
            }
            // merged scatter, one image row of corners at a time: walk the 4 pixels left to right with a pending
            // (address, value); a corner equal to the pending address is summed into it, anything else flushes.
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                int pa = -1;
                float pv = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int aL = half ? tp[r].i10 : tp[r].i00, aR = half ? tp[r].i11 : tp[r].i01;
                    const float vL = (half ? w10[r] : w00[r]) * gm[r], vR = (half ? w11[r] : w01[r]) * gm[r];
                    const bool live = tp[r].flags != 0;   // wholly invalid samples carry index 0: never touch it
                    if (live && aL == pa) {
                        pv += vL;
                    } else {
                        if (pa >= 0 && pv != 0.f) atomic_add_f32(L.gx + pa + c, pv);
                        pa = live ? aL : -1;
                        pv = vL;
                    }
                    if (live && aR == pa) {
                        pv += vR;
                    } else {
                        if (pa >= 0 && pv != 0.f) atomic_add_f32(L.gx + pa + c, pv);
                        pa = live ? aR : -1;
                        pv = vR;
                    }
                }
                if (pa >= 0 && pv != 0.f) atomic_add_f32(L.gx + pa + c, pv);
            }
        }
    };

    for (int t = 0; t < T; ++t) {
        it.next();
        const Chunk chn = it.get();   // chunk t + 1 (saturates at the last one)
        const bool tap_done = (t + 1 == T) || chn.k != ch.k || chn.dgi != ch.dgi;
        LSN_STAMP(2);
        // The slab of this chunk arrives by LDS-DMA (buffer_load ... lds): nothing orders a ds_read behind it except the
        // ISSUING wave's vmcnt followed by a barrier.  hipcc (ROCm 7.2) does put s_waitcnt vmcnt(0) in front of this
        // barrier while an LDS-DMA is in flight, but that is compiler behaviour, not a guarantee of the source: spell it.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // every wave's slab loads have landed
        LSN_STAMP(4);
        float xv0[4][4], xv1[4][4];
        if (want_off) issue_xv(ch, 0, xv0);
        // K = 256 here: one accumulator per tile (the two-level accumulation of the forward kernels pays from K ~ 1000)
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int so = (((s << 2) | kq) ^ j16) << 4;
            bf16x8 w0[NPL], w1[NPL];
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                w0[q] = *reinterpret_cast<const bf16x8 *>(b0 + q * PLANE + so);
                w1[q] = *reinterpret_cast<const bf16x8 *>(b1 + q * PLANE + so);
            }
#pragma unroll
            for (int prod = NP - 1; prod >= 0; --prod) {   // small terms first
                acc0 = mfma16_bf16(af[s][SC::pa(prod)], w0[SC::pb(prod)], acc0);
                acc1 = mfma16_bf16(af[s][SC::pa(prod)], w1[SC::pb(prod)], acc1);
            }
        }
        LSN_STAMP(5);
        __syncthreads();   // slab consumed by every wave: the next one may land
        if (t + 1 < T) issue_w(chn);
        if (want_off) issue_xv(ch, 1, xv1);
        LSN_STAMP(3);
        consume(ch, 0, acc0, xv0);
        consume(ch, 1, acc1, xv1);
        if (want_off && tap_done) {
            const int kd = ch.dgi * K + ch.k;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float h00 = row16_sum(H[r][0]), h01 = row16_sum(H[r][1]), h10 = row16_sum(H[r][2]),
                            h11 = row16_sum(H[r][3]);
                H[r][0] = H[r][1] = H[r][2] = H[r][3] = 0.f;
                if (j16 == 0) {
                    const Tap tp = tab[trow + r * KD + kd];
                    const float hy = 1.f - tp.ly, hx = 1.f - tp.lx;
                    const float v00 = (tp.flags & 1) ? h00 : 0.f, v01 = (tp.flags & 2) ? h01 : 0.f;
                    const float v10 = (tp.flags & 4) ? h10 : 0.f, v11 = (tp.flags & 8) ? h11 : 0.f;
                    // coordinate weights, kernel.cu:145-188 / 800-845, applied to the channel sums
                    const float dy = hx * (v10 - v00) + tp.lx * (v11 - v01);
                    const float dx = hy * (v01 - v00) + tp.ly * (v11 - v10);
                    const float bil = hy * hx * v00 + hy * tp.lx * v01 + tp.ly * hx * v10 + tp.ly * tp.lx * v11;
                    float *ga = gacc + (trow + r * KD + kd) * 3;
                    ga[0] += tp.m * dy;
                    ga[1] += tp.m * dx;
                    ga[2] += bil;
                }
            }
        }
        ch = chn;
        LSN_STAMP(6);
    }
    __syncthreads();

    for (int e = tid; e < BWD_BM * KD; e += 256) {
        const int pl = e / KD, r = e - pl * KD;
        const int dgi = r / K, k = r - dgi * K;
        const int pix = tile_p + pl;
        if (pix >= L.P || k < k_lo || k >= k_hi) continue;
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
            float gmv = ga[2];
            if (a.msig) {
                const float m = tab[e].m;
                gmv *= m * (1.f - m);
            }
            L.gmsk[(size_t)b * L.msb + (size_t)(dgi * K + k) * L.msc + (size_t)ho * L.msh + (size_t)wo * L.msw] = gmv;
        }
    }
}

// ---- the atomic-free gather path (anchor lists, per-anchor sums, combine, offset / mask gradients) ----
}  // namespace lsn
#include "dcn_gather_kernels.h"
namespace lsn {

// =============================================================================================
// Backward-weight:  gw[co][k,ci] += sum_p gout[p][co] * col[p][k,ci];  gb[co] += sum_p gout[p][co].
// GEMM with M = output channels (<=256 per block), N = 64 k-columns (one tap x 64 channels),
// reduction over pixels.  grid.x = column blocks, grid.y = pixel splits, grid.z = co blocks;
// each block walks its share of the 32-pixel steps of all levels, re-gathering col on the fly
// (the reference re-runs im2col for the same purpose, deform_conv_cuda.cpp:770-773), and ends
// with per-split partial gradients for the ordered reduce (a.wg_part) or, without them, fp32 atomics into gw.
// =============================================================================================
constexpr int WG_BP = 32, WG_BN = 64, WG_BM = 256;
template <bool VEC>
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
        const int rem = nco - gq * 4;
#pragma unroll
        for (int ps = 0; ps < NPA; ++ps) {
            const int pix = p0 + ps * 4 + grow;
            const bool ok = pix < L.P;
            gv[ps] = load4_guarded<VEC>(L.gout + (size_t)(ok ? pix : 0) * a.Co + co_base, gq * 4, rem, ok);
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

    // a.wg_part (round 6): per-split partial gradients + the ordered reduce instead of fp32 atomics: the exact mode's weight
    // gradients are bit-reproducible like the default mode's (a split without steps still stores its zero tile)
    float *pw = a.wg_part ? a.wg_part + (size_t)blockIdx.y * a.Co * Kdim : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = j * 32 + (lane & 31);
            if (col >= ch.nval) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wave * 64 + i * 32 + mfma32_row(r, lane);
                if (row >= nco) continue;
                const size_t e = (size_t)(co_base + row) * Kdim + ch.k * Cg + ch.c0 + col;
                if (pw)
                    pw[e] = acc[i][j][r];   // this pixel split's partial gradient: the ordered reduce adds the splits
                else
                    atomic_add_f32(a.gw + e, acc[i][j][r]);
            }
        }
    if (do_bias && tid < nco) {
        if (a.wg_part_b)
            a.wg_part_b[(size_t)blockIdx.y * a.Co + co_base + tid] = bias_acc;
        else
            atomic_add_f32(a.gb + co_base + tid, bias_acc);
    }
}

// =============================================================================================
// Backward-weight on the bf16 matrix pipe with split operands (dcn_wgrad_xn_kernel).
//
// Same decomposition as dcn_wgrad_kernel (grid = column blocks x pixel splits x co blocks, 32-pixel steps, fp32 atomics
// into gw at the end).  The reduction index of this GEMM is the PIXEL, so both LDS images are built pixel-contiguous
// -- As[co][32 px], Bs[kcol][32 px], NPL bf16 planes each (2 or 3), 80-byte rows -- to give every lane the 8
// consecutive k-values v_mfma_f32_32x32x16_bf16 wants.  A thread owns two adjacent output channels x 16 pixels of
// grad_output and two adjacent channels x 4 pixels of the gathered columns (8-byte buffer loads, out-of-range rows read
// as zero), splits its values in registers and writes 16- resp. 8-byte row pieces.  48 (24) bf16 MFMAs per step and wave.
// The step's sampling table comes from the launch-wide table of the backward-data pass when there is one (a.gtap).
// PLAIN: a dense convolution (no offsets, no mask): one load per sample instead of four corners.
// =============================================================================================
template <int NP, int BMW = WG_BM>
__host__ __device__ inline size_t wgrad_xn_lds_bytes()
{
    return (size_t)SplitCfg<NP>::NPL * (BMW + WG_BN) * 80 + 3 * WG_BP * sizeof(Tap);
}

// BMW: output channels per block.  256 (four waves x 64 co x 64 columns) for the wide layers; 64 (four waves x 32 x 32)
// for convolutions with few output channels (a DCNv2 pack's 27-channel offset conv, the 64-channel stem stage), whose
// weight gradient would otherwise spend 3/4 .. 9/10 of its MFMAs and of its gout staging on rows that do not exist.
template <bool PLAIN, int NP, int BMW = WG_BM>
__global__ __launch_bounds__(256, 2) void dcn_wgrad_xn_kernel(const DcnArgs a, int nsteps)
{
    static_assert(BMW == 256 || BMW == 64, "co tile");
    constexpr int TI = BMW == 256 ? 2 : 1;   // 32x32 accumulator tiles per wave: TI x TI
    using SC = SplitCfg<NP>;
    constexpr int NPL = SC::NPL;
    constexpr int RS = 80, PLANE_A = BMW * RS, PLANE_B = WG_BN * RS;
    extern __shared__ __align__(16) unsigned char smem[];   // [A planes][B planes][tab 3 x 32]
    Tap *tab = reinterpret_cast<Tap *>(smem + NPL * PLANE_A + NPL * PLANE_B);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.kh * a.kw;
    const int Cg = a.C / a.groups, Cog = a.Co / a.groups, Kdim = K * Cg;
    const int segs = Cg / a.SL, ncc = (a.SL + WG_BN - 1) / WG_BN;
    const int ncol_g = K * segs * ncc;
    // work item = (pixel split, column block), XCD-ordered with the column blocks of one split adjacent: they
    // read the same gout rows and the same input rows
    const int ncol_all = gridDim.x, nsplit = gridDim.y;
    const int work = xcd_remap(blockIdx.y * ncol_all + blockIdx.x, ncol_all * nsplit);
    const int bsplit = work / ncol_all, bcol = work - bsplit * ncol_all;
    const int g = bcol / ncol_g;
    const Chunk ch = decode_chunk<WG_BN>(a, g, bcol - g * ncol_g, segs, ncc);
    const int co_blk = blockIdx.z * BMW;
    const int nco = min(BMW, Cog - co_blk);
    const int co_base = g * Cog + co_blk;
    const int row0 = BMW == 256 ? wave * 64 : (wave >> 1) * 32, col0 = BMW == 256 ? 0 : (wave & 1) * 32;

    const int st_begin = (int)((long long)nsteps * bsplit / nsplit);
    const int st_end = (int)((long long)nsteps * (bsplit + 1) / nsplit);

    // gathered columns: a thread owns two adjacent channels (one 8-byte load per corner) of a 4-pixel group
    const int kp = tid & 31, pg = tid >> 5;
    const bool cval0 = 2 * kp < ch.nval, cval1 = 2 * kp + 1 < ch.nval;
    const int cbase = g * Cg + ch.c0;
    // gout: a thread owns two adjacent output channels (one 8-byte load per pixel) and one half of the step's pixels
    constexpr int CP = BMW / 2;
    const int gcp = tid % CP, gph = tid / CP;   // co pair, pixel half; threads >= 2 CP (BMW = 64: waves 1..3) idle here
    const bool gact = tid < 2 * CP;
    const bool gval0 = gact && 2 * gcp < nco, gval1 = gact && 2 * gcp + 1 < nco;
    const bool do_bias = (a.gb != nullptr) && ch.k == 0 && ch.c0 == 0;
    const int kd = ch.dgi * K + ch.k;

    // 8-byte buffer loads need even channel counts and bases, 8-byte aligned tensors and byte offsets below 2^31
    // (tensor alignment and sizes: checked by the host, a.wg_vec bit 0 = input, bit 1 = grad_output)
    bool vx = (a.wg_vec & 1) && (cbase & 1) == 0, vg = (a.wg_vec & 2) && (co_base & 1) == 0;
    if ((a.dbg_block >> 25) & 1) vx = vg = false;   // diagnostic: scalar loads
    // the backward-data pass of the same call left the sampling table of every (pixel, tap) in a.gtap (k-major)
    const bool use_gtap = !PLAIN && vx && vg && a.gtap != nullptr && !((a.dbg_block >> 24) & 1);

    constexpr int NX = PLAIN ? 1 : 4;
    float xv0[4][NX], xv1[4][NX];
    float gv0[WG_BP / 2], gv1[WG_BP / 2];
    float bias_acc0 = 0.f, bias_acc1 = 0.f;
    int npx_s = 0;   // scalar gout path: valid pixels of this thread's half in the step whose values sit in gv0 / gv1

    f32x16 acc[TI][TI];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // computed table entry of pixel `tid` of step st (no launch-wide table)
    auto tap_of = [&](int st) __attribute__((always_inline)) -> Tap {
        const Lvl &L = find_level(a, st);
        return make_tap(a, L, (st - L.tile0) * WG_BP + tid, ch.k, ch.dgi);
    };
    // launch-wide table: the step's 32 entries are 1 KB contiguous; lane l of wave 0 moves 16-byte half (l & 1) of
    // entry (l >> 1).  Pixels past the level's end take the all-zero entry behind the table ("no sample") -- by
    // address, not by a select on the loaded value, which would make the wave wait for the load where it is issued.
    auto gtap_load = [&](int st) __attribute__((always_inline)) -> uint4 {
        const Lvl &L = find_level(a, st);
        const int pix = (st - L.tile0) * WG_BP + (lane >> 1);
        const size_t ent = pix < L.P ? (size_t)kd * a.gtap_rows + L.prow0 + pix : (size_t)K * a.dg * a.gtap_rows;
        return reinterpret_cast<const uint4 *>(a.gtap)[ent * 2 + (lane & 1)];
    };
    auto gtap_put = [&](int slot, uint4 v) __attribute__((always_inline)) { reinterpret_cast<uint4 *>(tab + slot * WG_BP)[lane] = v; };

    // issue the global loads of step st (table slot buf); VX / VG: 8-byte buffer loads, out-of-range rows read as 0
    auto load_step = [&](int st, int buf, auto vx_, auto vg_) __attribute__((always_inline)) {
        constexpr bool VX = decltype(vx_)::value, VG = decltype(vg_)::value;
        const Lvl &L = find_level(a, st);
        const int p0 = (st - L.tile0) * WG_BP;
        if constexpr (VX) {
            const __amdgpu_buffer_rsrc_t xrs =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(L.x), 0, L.B * L.H * L.W * a.C * 4, 0x00020000);
            const int c4 = (cbase + 2 * kp) * 4;   // (columns past nval hold other channels' values: never written back)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const Tap *tp = &tab[buf * WG_BP + pg * 4 + q];
                int idx[4];
                if (PLAIN) {
                    idx[0] = tp->i00;
                } else {
                    const int4 i4 = *reinterpret_cast<const int4 *>(tp);
                    idx[0] = i4.x, idx[1] = i4.y, idx[2] = i4.z, idx[3] = i4.w;
                }
#pragma unroll
                for (int e = 0; e < NX; ++e) {
                    const float2 v = buf_load_f32x2(xrs, idx[e] * 4 + c4, 0);
                    xv0[q][e] = v.x, xv1[q][e] = v.y;
                }
            }
        } else {
            const int c = cbase + (cval0 ? 2 * kp : 0), c1 = cval1 ? 1 : 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const Tap *tp = &tab[buf * WG_BP + pg * 4 + q];
                int idx[4];
                if (PLAIN) {
                    idx[0] = tp->i00;
                } else {
                    const int4 i4 = *reinterpret_cast<const int4 *>(tp);
                    idx[0] = i4.x, idx[1] = i4.y, idx[2] = i4.z, idx[3] = i4.w;
                }
#pragma unroll
                for (int e = 0; e < NX; ++e) {
                    const float *xp = L.x + idx[e] + c;
                    xv0[q][e] = xp[0], xv1[q][e] = xp[c1];
                }
            }
        }
        if (gact) {   // (wave-uniform: BMW = 64 is the first wave)
            const int pbase = gph * (WG_BP / 2);
            if constexpr (VG) {
                const __amdgpu_buffer_rsrc_t grs =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(L.gout), 0, L.P * a.Co * 4, 0x00020000);
                const int v0 = ((p0 + pbase) * a.Co + co_base + 2 * gcp) * 4, rowb = a.Co * 4;
#pragma unroll
                for (int px = 0; px < WG_BP / 2; ++px) {
                    const float2 v = buf_load_f32x2(grs, v0 + px * rowb, 0);
                    gv0[px] = v.x, gv1[px] = v.y;
                }
            } else {
                const int npx = L.P - p0 - pbase;   // valid pixels of this half (may be <= 0); masked when staged
                npx_s = npx;
                const float *gp = L.gout + (size_t)p0 * a.Co + co_base + (gval0 ? 2 * gcp : 0);
                const int g1 = gval1 ? 1 : 0;
#pragma unroll
                for (int px = 0; px < WG_BP / 2; ++px) {
                    const float *q = gp + (size_t)(px < npx ? pbase + px : 0) * a.Co;
                    gv0[px] = q[0], gv1[px] = q[g1];
                }
            }
        }
    };
    // split the values loaded by load_step into bf16 planes and write the two LDS operand images
    auto store_step = [&](int buf, auto vx_, auto vg_) __attribute__((always_inline)) {
        constexpr bool VX = decltype(vx_)::value, VG = decltype(vg_)::value;
        // gathered columns: 4 pixels of channels 2 kp, 2 kp + 1 -> one 8-byte piece of rows 2 kp (+1) in each plane
        float v0[4], v1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const Tap tp = tab[buf * WG_BP + pg * 4 + q];
            if constexpr (PLAIN) {
                v0[q] = (tp.flags & 1) ? xv0[q][0] : 0.f;
                v1[q] = (tp.flags & 1) ? xv1[q][0] : 0.f;
            } else {
                float b00, b01, b10, b11;
                corner_weights(tp, b00, b01, b10, b11);
                v0[q] = (b00 * xv0[q][0] + b01 * xv0[q][1] + b10 * xv0[q][2] + b11 * xv0[q][3]) * tp.m;
                v1[q] = (b00 * xv1[q][0] + b01 * xv1[q][1] + b10 * xv1[q][2] + b11 * xv1[q][3]) * tp.m;
            }
            if (!VX) {
                v0[q] = cval0 ? v0[q] : 0.f;
                v1[q] = cval1 ? v1[q] : 0.f;
            }
        }
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
            unsigned cp[2][NPL];
            split_planes<NPL>(e2 ? v1[0] : v0[0], e2 ? v1[1] : v0[1], cp[0]);
            split_planes<NPL>(e2 ? v1[2] : v0[2], e2 ? v1[3] : v0[3], cp[1]);
            unsigned char *bp = smem + NPL * PLANE_A + (2 * kp + e2) * RS + pg * 8;
#pragma unroll
            for (int q = 0; q < NPL; ++q) *reinterpret_cast<uint2 *>(bp + q * PLANE_B) = make_uint2(cp[0][q], cp[1][q]);
        }
        // gout: 16 pixels of output channels 2 gcp, 2 gcp + 1 -> two 16-byte pieces of rows 2 gcp (+1) in each plane.
        // (VG: pixels past the level's end were read as 0; rows past nco hold other values but are never written back)
        if (gact) {
            if (!VG) {
#pragma unroll
                for (int px = 0; px < WG_BP / 2; ++px) {
                    gv0[px] = (gval0 && px < npx_s) ? gv0[px] : 0.f;
                    gv1[px] = (gval1 && px < npx_s) ? gv1[px] : 0.f;
                }
            }
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2)
#pragma unroll
                for (int piece = 0; piece < 2; ++piece) {
                    unsigned gp4[4][NPL];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float w0 = e2 ? gv1[piece * 8 + 2 * e] : gv0[piece * 8 + 2 * e];
                        const float w1 = e2 ? gv1[piece * 8 + 2 * e + 1] : gv0[piece * 8 + 2 * e + 1];
                        split_planes<NPL>(w0, w1, gp4[e]);
                    }
                    unsigned char *ap = smem + (2 * gcp + e2) * RS + gph * 32 + piece * 16;
#pragma unroll
                    for (int q = 0; q < NPL; ++q)
                        *reinterpret_cast<uint4 *>(ap + q * PLANE_A) = make_uint4(gp4[0][q], gp4[1][q], gp4[2][q], gp4[3][q]);
                }
            if (do_bias) {
#pragma unroll
                for (int px = 0; px < WG_BP / 2; ++px) bias_acc0 += gv0[px], bias_acc1 += gv1[px];
            }
        }
    };

#ifdef LSN_WG_INTERLEAVE
    // Experiment (not the default build): the 8-byte loads of step st + 1 issued one by one in the gaps between the MFMAs
    // of step st instead of as a block before them (the block costs ~2.5 k cycles of pure issue time per step).
    constexpr int IL_NL = 4 * NX + WG_BP / 2, IL_NG = 2 * NP * TI * TI;
    int il_xoff[4][NX];
    int il_g0 = 0, il_rowb = 0, il_xbytes = 0, il_gbytes = 0;
    const float *il_xp = nullptr, *il_gp = nullptr;
    auto il_prepare = [&](int st, int buf) __attribute__((always_inline)) {
        const Lvl &L = find_level(a, st);
        const int p0 = (st - L.tile0) * WG_BP;
        il_xp = L.x, il_gp = L.gout;
        il_xbytes = L.B * L.H * L.W * a.C * 4, il_gbytes = L.P * a.Co * 4;
        const int c4 = (cbase + 2 * kp) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const Tap *tp = &tab[buf * WG_BP + pg * 4 + q];
            if (PLAIN) {
                il_xoff[q][0] = tp->i00 * 4 + c4;
            } else {
                const int4 i4 = *reinterpret_cast<const int4 *>(tp);
                il_xoff[q][0] = i4.x * 4 + c4, il_xoff[q][NX > 1 ? 1 : 0] = i4.y * 4 + c4;
                il_xoff[q][NX > 2 ? 2 : 0] = i4.z * 4 + c4, il_xoff[q][NX > 3 ? 3 : 0] = i4.w * 4 + c4;
            }
        }
        il_g0 = ((p0 + gph * (WG_BP / 2)) * a.Co + co_base + 2 * gcp) * 4;
        il_rowb = a.Co * 4;
    };
    auto il_issue = [&](int m) __attribute__((always_inline)) {
        if (m < 4 * NX) {
            const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(il_xp), 0, il_xbytes, 0x00020000);
            const int q = m / NX, e = m - q * NX;
            const float2 v = buf_load_f32x2(xrs, il_xoff[q][e], 0);
            xv0[q][e] = v.x, xv1[q][e] = v.y;
        } else if (gact) {
            const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(il_gp), 0, il_gbytes, 0x00020000);
            const int px = m - 4 * NX;
            const float2 v = buf_load_f32x2(grs, il_g0 + px * il_rowb, 0);
            gv0[px] = v.x, gv1[px] = v.y;
        }
    };
#endif
    int dbg_n = 0;
    // Tap-table slots rotate over three steps: slot (i % 3) holds step st_begin + i.  Iteration st stages step st,
    // issues the loads of step st + 1 (indices from slot st + 1) and, after its MFMA block, fills slot st + 2:
    //   GT  (launch-wide table): wave 0 writes the 1 KB it loaded one iteration earlier and issues the load for
    //       step st + 3 -- a full iteration of slack for an HBM miss, four registers;
    //   !GT (computed): the entry's offset / mask loads are issued before the MFMA block and finished after it.
    // (GT: the table load is issued by every wave, unconditionally and always as the YOUNGEST load in flight, also in
    // the prologue: only then can the compiler count on it and let the staging code wait with vmcnt(1) instead of 0.)
#ifdef LSN_WG_INTERLEAVE
#define LSN_WG_LOADS(st, slot1, VXT, VGT)                                                                             \
    do {                                                                                                              \
        if constexpr (VXT::value && VGT::value) il_prepare(min(st + 1, st_end - 1), (st + 1 < st_end) ? slot1 : slot); \
        else if (st + 1 < st_end) load_step(st + 1, slot1, VXT{}, VGT{});                                             \
    } while (0)
#define LSN_WG_GAP(gap, VXT, VGT)                                                                                     \
    do {                                                                                                              \
        if constexpr (VXT::value && VGT::value) {                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
_Pragma("unroll")                                                                                                     \
            for (int m_ = (gap) * IL_NL / IL_NG; m_ < ((gap) + 1) * IL_NL / IL_NG; ++m_) il_issue(m_);                 \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
        }                                                                                                             \
    } while (0)
#else
#define LSN_WG_LOADS(st, slot1, VXT, VGT)                                                                             \
    do {                                                                                                              \
        if (st + 1 < st_end) load_step(st + 1, slot1, VXT{}, VGT{});                                                  \
    } while (0)
#define LSN_WG_GAP(gap, VXT, VGT)                                                                                     \
    do {                                                                                                              \
    } while (0)
#endif
#define LSN_WG_RUN(VXT, VGT, GTB)                                                                                     \
    do {                                                                                                              \
        uint4 tq = make_uint4(0u, 0u, 0u, 0u);                                                                        \
        if constexpr (GTB) {                                                                                          \
            if (wave == 0) {                                                                                          \
                gtap_put(0, gtap_load(st_begin));                                                                     \
                if (st_begin + 1 < st_end) gtap_put(1, gtap_load(st_begin + 1));                                      \
            }                                                                                                         \
        } else {                                                                                                      \
            if (tid < WG_BP) {                                                                                        \
                tab[tid] = tap_of(st_begin);                                                                          \
                if (st_begin + 1 < st_end) tab[WG_BP + tid] = tap_of(st_begin + 1);                                   \
            }                                                                                                         \
        }                                                                                                             \
        __syncthreads();                                                                                              \
        load_step(st_begin, 0, VXT{}, VGT{});                                                                         \
        if constexpr (GTB) tq = gtap_load(min(st_begin + 2, st_end - 1));                                             \
        for (int st = st_begin; st < st_end; ++st) {                                                                  \
            const int slot = (st - st_begin) % 3, slot1 = (slot + 1) % 3, slot2 = (slot + 2) % 3;                     \
            LSN_STAMP(2);                                                                                             \
            store_step(slot, VXT{}, VGT{});                                                                           \
            LSN_STAMP(3);                                                                                             \
            __syncthreads();                                                                                          \
            LSN_STAMP(4);                                                                                             \
            const bool build2 = !(GTB) && st + 2 < st_end && tid < WG_BP;                                             \
            Tap t2 = {};                                                                                              \
            if (build2) t2 = tap_of(st + 2);                                                                          \
            LSN_WG_LOADS(st, slot1, VXT, VGT);                                                                        \
            LSN_STAMP(5);                                                                                             \
            const unsigned char *ap = smem + (row0 + (lane & 31)) * RS + (lane >> 5) * 16;                            \
            const unsigned char *bp = smem + NPL * PLANE_A + (col0 + (lane & 31)) * RS + (lane >> 5) * 16;            \
_Pragma("unroll")                                                                                                     \
            for (int ks = 0; ks < 2; ++ks) {                                                                          \
                bf16x8 Af[TI][NPL], Bf[TI][NPL];                                                                      \
_Pragma("unroll")                                                                                                     \
                for (int i = 0; i < TI; ++i)                                                                          \
_Pragma("unroll")                                                                                                     \
                    for (int q = 0; q < NPL; ++q) {                                                                   \
                        Af[i][q] = *reinterpret_cast<const bf16x8 *>(ap + q * PLANE_A + i * 32 * RS + ks * 32);       \
                        Bf[i][q] = *reinterpret_cast<const bf16x8 *>(bp + q * PLANE_B + i * 32 * RS + ks * 32);       \
                    }                                                                                                 \
_Pragma("unroll")                                                                                                     \
                for (int prod = 0; prod < NP; ++prod)                                                                 \
_Pragma("unroll")                                                                                                     \
                    for (int i = 0; i < TI; ++i)                                                                      \
_Pragma("unroll")                                                                                                     \
                        for (int j = 0; j < TI; ++j)                                                                  \
                        {                                                                                             \
                            acc[i][j] = mfma_bf16(Af[i][SC::pa(prod)], Bf[j][SC::pb(prod)], acc[i][j]);               \
                            LSN_WG_GAP(((ks * NP + prod) * TI + i) * TI + j, VXT, VGT);                               \
                        }                                                                                             \
            }                                                                                                         \
            LSN_STAMP(6);                                                                                             \
            if constexpr (GTB) {                                                                                      \
                if (wave == 0 && st + 2 < st_end) gtap_put(slot2, tq);                                                \
                tq = gtap_load(min(st + 3, st_end - 1));                                                              \
            } else {                                                                                                  \
                if (build2) tab[slot2 * WG_BP + tid] = t2;                                                            \
            }                                                                                                         \
            __syncthreads();                                                                                          \
            LSN_STAMP(7);                                                                                             \
        }                                                                                                             \
    } while (0)
    if (st_begin < st_end) {
        using T = std::true_type;
        using F = std::false_type;
        if (use_gtap) {
            if constexpr (!PLAIN) LSN_WG_RUN(T, T, true);
        } else if (vx && vg) {
            LSN_WG_RUN(T, T, false);
        } else {
            // (Two scalar variants -- input only, both -- pushed the uses of `a` past what the compiler will trace
            // when it turns the by-value argument copy into kernarg loads: the whole struct landed in scratch.
            // Check ScratchSize with -Rpass-analysis=kernel-resource-usage after touching this kernel.)
            LSN_WG_RUN(F, F, false);
        }
    }
#undef LSN_WG_RUN
#undef LSN_WG_LOADS
#undef LSN_WG_GAP

    // every element of the gradient belongs to exactly one (column block, co block): with a partial buffer per pixel
    // split the epilogue is a plain store and the splits are added in a fixed order afterwards
    float *pw = a.wg_part ? a.wg_part + (size_t)bsplit * a.Co * Kdim : nullptr;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TI; ++j) {
            const int col = col0 + j * 32 + (lane & 31);
            if (col >= ch.nval) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + mfma32_row(r, lane);
                if (row >= nco) continue;
                const size_t e = (size_t)(co_base + row) * Kdim + ch.k * Cg + ch.c0 + col;
                if (pw)
                    pw[e] = acc[i][j][r];
                else
                    atomic_add_f32(a.gw + e, acc[i][j][r]);
            }
        }
    if (a.wg_part_b) {   // (kernel-argument condition: every thread takes the barriers)
        // a channel pair's sum sits in two threads (one per half of a step's pixels): meet in LDS, store once
        __syncthreads();
        float *bred = reinterpret_cast<float *>(smem);
        if (gact) bred[(gph * CP + gcp) * 2] = bias_acc0, bred[(gph * CP + gcp) * 2 + 1] = bias_acc1;
        __syncthreads();
        if (do_bias && gact && gph == 0) {
            const float s0 = bred[gcp * 2] + bred[(CP + gcp) * 2], s1 = bred[gcp * 2 + 1] + bred[(CP + gcp) * 2 + 1];
            if (gval0) a.wg_part_b[(size_t)bsplit * a.Co + co_base + 2 * gcp] = s0;
            if (gval1) a.wg_part_b[(size_t)bsplit * a.Co + co_base + 2 * gcp + 1] = s1;
        }
    } else {
        if (do_bias && gval0) atomic_add_f32(a.gb + co_base + 2 * gcp, bias_acc0);
        if (do_bias && gval1) atomic_add_f32(a.gb + co_base + 2 * gcp + 1, bias_acc1);
    }
}

}  // namespace lsn
