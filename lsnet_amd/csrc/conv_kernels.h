// Dense-convolution kernels of liblsnet_hip.so (included by conv.hip): implicit GEMM on the bf16 matrix pipe with split
// fp32 operands (common.h), designed for TWO workgroups per CU.
//
//   out[p][co] = sum_{tap, ci} x[p @ tap][ci] * w[co][tap][ci]            (channels-last fp32 tensors)
//
// * The WEIGHTS never pass through LDS.  conv_wfrag_kernel writes them once per optimizer step in MFMA FRAGMENT ORDER
//   ([chunk][32-co tile][k-step][plane][lane][8 bf16]: the 1 KiB a wave needs for one operand of one MFMA is one
//   contiguous, fully coalesced buffer_load_dwordx4), so a wave fetches its own weight fragments from L2 straight into
//   registers, half a chunk ahead of their use.  No staging instructions, no LDS space, no barrier for that operand.
// * The PIXELS (fp32 in HBM) are loaded as float4 per lane (8 lanes = one pixel's 32-channel slab = 128 B), split
//   exactly into bf16 planes in registers and written to a double-buffered LDS image of 64-byte rows whose 16-byte slots
//   are XOR-swizzled by (row >> 2) & 3: the ds_write_b64 stores and the ds_read_b128 fragment reads are both
//   conflict-free without padding.  24 KB per stage for 128 pixels => 48 KB per workgroup.
// * Operand roles are SWAPPED in the MFMA (rows of D = output channels, columns = pixels): a lane then owns ONE pixel of
//   a 32x32 tile and four consecutive output channels per register quad, so the epilogue is one 16-byte store per quad
//   (bias and ReLU applied on the way) instead of sixteen 4-byte stores, and the pixel's output address is computed once.
// * <= 256 VGPRs and 48 KB of LDS: two workgroups share a CU, i.e. two waves per SIMD that are NOT barrier-coupled --
//   one's split / staging VALU work and barrier waits sit in the shadow of the other's MFMAs, tiles are handed out at
//   512 slots per round instead of 256, and twice the bytes are in flight on the layers that are bound by HBM latency.
//
// Template: wave tile = (TM x 32 pixels) x (TN x 32 output channels), workgroup = WM x WN waves (= 4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace lsn {

constexpr int CV_MAXLV = 16;   // (the pyramid deformable op batches 15 (level, source) pairs)

// One input map of a batched launch: the FPN levels that share a convolution's weights (LSHead) go into ONE launch.
struct ConvLvl {
    const float *x;
    const float *res;   // optional residual of the output's shape, added before the ReLU (may alias `out`: accumulate)
    const float *gate;  // optional tensor of the output's shape: the finished element is zeroed where gate <= 0 (the ReLU
                        // gate of the activation whose gradient this launch produces, fused into a backward-data pass)
    float *out;
    int B, H, W, Ho, Wo;
    int P;       // B * Ho * Wo
    int tile0;   // first pixel tile of this level
};

struct ConvArgs {
    ConvLvl lv[CV_MAXLV];
    int nlv, ntiles;
    const float *bias;
    int C, Co, kh, kw, stride, pad_h, pad_w, dil;
    int xpitch;   // floats between horizontally adjacent input pixels (= C, except for the row-merged stem form)
    int relu;
    int ksplit;   // > 1 (one level, dense output): the chunk range is divided over blockIdx.z and split z stores its
                  // partial tile at part + z * P * Co; conv_splitk_reduce_kernel adds them up (+ bias, ReLU).  Few
                  // pixels under a deep reduction (layer 4, FPN P5 .. P7) would otherwise leave most CUs idle; fp32
                  // atomics into the output were measured 5 - 10 x slower than the whole convolution.
    float *part;
    int tile_base;              // first pixel tile of this launch (a launch may cover a tile range of the level: conv.hip tail split)
    int part_pix0, part_rows;   // partial tiles: row r of split z lies at part + (z * part_rows + r - part_pix0) * Co
    const unsigned short *wf;   // weights in fragment order (conv_wfrag_kernel)
    int wf_bytes;
    // output placement: pixel (b, ho, wo) of the (Ho, Wo) grid is stored at (b, oy0 + ho * ostep, ox0 + wo * ostep) of an
    // (OH, OW) map.  ostep = 0: the dense case.  Used by the strided backward-data pass (one residue class per launch).
    int ostep, oy0, ox0, OH, OW;
    // ---- work distribution (conv.hip sk_plan).  The grid is one-dimensional in x: workgroups [0, n_dp) take one whole
    // (pixel tile, column block) each; workgroups [n_dp, n_dp + sk_n) share the remaining sk_tiles tiles EVENLY BY CHUNKS
    // ("stream-K"): piece s owns the chunk units [U s / sk_n, U (s + 1) / sk_n) of the tile-major unit space, U = sk_tiles *
    // (chunks per tile).  A piece that covers only part of a tile leaves its accumulators in slot 2 s (the segment that
    // opens the piece) or 2 s + 1 (the one that closes it) of sk_part and takes a ticket of the tile's counter; the piece
    // that draws the LAST ticket adds the tile's slots in chunk order -- a fixed order, whoever arrives last: the same bits
    // on every run, no second launch -- runs the epilogue and re-arms the counter.  Nobody waits for anybody.
    int colblocks, n_dp, sk_n, sk_tiles;
    float *sk_part;
    unsigned *sk_cnt;   // one counter per stream-K tile, zero between launches
};

// Tap subset of a transposed convolution: taps i = i0 + m * istep (m < ni), j likewise.
struct TapSub {
    int i0, istep, ni, j0, jstep, nj, kw;
};

__host__ __device__ inline int cv_ncc(int C) { return (C + 31) / 32; }
// 32-channel output tiles of the weight image, padded with zero tiles to the width of the workgroup tile that serves
// this Co (conv.hip conv_forward: 32, 64 or 128 columns): the scalar part of a buffer address is not range-checked, so
// every tile a wave may ask for has to exist.
__host__ __device__ inline int cv_nt(int Co) { return Co <= 32 ? 1 : Co <= 64 ? 2 : (Co + 127) / 128 * 4; }
// bytes of the fragment-order image of a (Co, Kd, C) weight
__host__ __device__ inline size_t cv_wfrag_bytes(int Co, int Kd, int C, int npl)
{
    return (size_t)Kd * cv_ncc(C) * cv_nt(Co) * 2 * npl * 1024;
}

// One weight -> fragment order.  GEMM view of the convolution the main kernel runs: output column n (< Co), reduction
// index (tap, c) with c < C.  flipT = 0: Wt[n][tap][c] = w[(n * K + tap) * C + c].  flipT = 1 (backward-data): the source is
// the forward weight (this GEMM's C = forward Co), Wt[n][tap'][c] = w[(c * K + tap(tap')) * Co + n] with the tap subset
// reversed.  Element (t, nt, ks, q, lane, e): n = nt * 32 + (lane & 31), k = ks * 16 + 8 * (lane >> 5) + e, value = plane q
// of Wt[n][tap(t)][cc(t) * 32 + k] (0 beyond Co / C).
// Optional per-output-channel scale of an eval-mode BatchNorm folded into the convolution (bn_gamma != NULL): every
// weight of FORWARD output channel co is multiplied by gamma[co] / sqrt(var[co] + eps) on the way, and (shift_out != NULL)
// shift[co] = beta[co] - mean[co] * that scale is written for the convolution's bias slot.
struct WfragJob {
    const float *w;
    unsigned short *out;
    int Co, K, C, flipT;
    TapSub ts;
    long long start;     // first thread id of this job in a multi-job launch
    const float *bn_gamma, *bn_var, *bn_beta, *bn_mean;
    float *shift_out;
    float bn_eps;
};

template <int NPL>
__device__ __forceinline__ void wfrag_item(const WfragJob &jb, long long id)
{
    const float *__restrict__ w = jb.w;
    const int Co = jb.Co, K = jb.K, C = jb.C, flipT = jb.flipT;
    const TapSub &ts = jb.ts;
    if (jb.shift_out != nullptr && id < (flipT ? C : Co)) {   // (a launch has at least 128 threads per 32 output columns)
        const int co = (int)id;
        jb.shift_out[co] = jb.bn_beta[co] - jb.bn_mean[co] * (jb.bn_gamma[co] * rsqrtf(jb.bn_var[co] + jb.bn_eps));
    }
    const int ncc = cv_ncc(C), NT = cv_nt(Co);
    const int lane = (int)(id & 63);
    long long r = id >> 6;
    const int ks = (int)(r & 1);
    r >>= 1;
    const int nt = (int)(r % NT);
    const int t = (int)(r / NT);
    const int tap = t / ncc, cc = t - tap * ncc;
    const int n = nt * 32 + (lane & 31);
    const int c0 = cc * 32 + ks * 16 + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = c0 + e;
        float val = 0.f;
        if (n < Co && c < C) {
            int cof;   // forward output channel of this weight
            if (!flipT) {
                val = w[((size_t)n * K + tap) * C + c];
                cof = n;
            } else {
                // transposed convolution: this GEMM's (n, c) = forward (ci, co); source layout (Co_f = C, K, C_f = Co)
                const int m = tap / ts.nj, nn = tap - m * ts.nj;
                const int i = ts.i0 + (ts.ni - 1 - m) * ts.istep, j = ts.j0 + (ts.nj - 1 - nn) * ts.jstep;
                val = w[((size_t)c * K + i * ts.kw + j) * Co + n];
                cof = c;
            }
            if (jb.bn_gamma != nullptr) val *= jb.bn_gamma[cof] * rsqrtf(jb.bn_var[cof] + jb.bn_eps);
        }
        v[e] = val;
    }
    unsigned pl[4][NPL];
#pragma unroll
    for (int e = 0; e < 4; ++e) split_planes<NPL>(v[2 * e], v[2 * e + 1], pl[e]);
    unsigned short *dst = jb.out + ((((size_t)t * NT + nt) * 2 + ks) * NPL) * 512 + (size_t)lane * 8;
#pragma unroll
    for (int q = 0; q < NPL; ++q)
        *reinterpret_cast<uint4 *>(dst + (size_t)q * 512) = make_uint4(pl[0][q], pl[1][q], pl[2][q], pl[3][q]);
}

__host__ __device__ inline long long wfrag_threads(const WfragJob &jb)
{
    const int Kd = jb.flipT ? jb.ts.ni * jb.ts.nj : jb.K;
    return (long long)Kd * cv_ncc(jb.C) * cv_nt(jb.Co) * 2 * 64;   // one thread per (t, nt, ks, lane)
}

template <int NPL>
__global__ void conv_wfrag_kernel(const WfragJob jb)
{
    const long long total = wfrag_threads(jb);
    for (long long id = blockIdx.x * (long long)blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x)
        wfrag_item<NPL>(jb, id);
}

// Many weights in ONE launch (the images of every trainable convolution after an optimizer step): job j owns the
// thread ids [start_j, start_{j+1}).
template <int NPL>
__global__ void conv_wfrag_multi_kernel(const WfragJob *__restrict__ jobs, int njobs, long long total)
{
    for (long long id = blockIdx.x * (long long)blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        // a job's thread count is a multiple of 128 (wfrag_threads), so the 64 consecutive ids of a wave share their job:
        // the search runs on the wave's first id, in scalar registers (per-lane it was eight dependent vector loads)
        const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)(id & 0xffffffffll));
        const unsigned hi32 = __builtin_amdgcn_readfirstlane((unsigned)(id >> 32));
        const long long id0 = ((long long)hi32 << 32) | lo32;
        int lo = 0, hi = njobs - 1;   // last job with start <= id0
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].start <= id0) lo = mid; else hi = mid - 1;
        }
        const WfragJob jb = jobs[lo];
        wfrag_item<NPL>(jb, id - jb.start);
    }
}

// ds_read_b64_tr_b16: within a 16-lane group lane i supplies the address of 8-byte chunk i, chunks 4r .. 4r+3 are row r,
// and lane i receives [row 0..3][column i] (tools/ubench/tr16_probe.hip)
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4 lds_tr16(const unsigned char *p)
{
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3))) *)((__attribute__((address_space(3))) const unsigned char *)p));
}

__device__ __forceinline__ float4 cv_load4(__amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    float4 f;
    __builtin_memcpy(&f, &v, 16);
    return f;
}
__device__ __forceinline__ bf16x8 cv_load_frag(__amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    bf16x8 f;
    __builtin_memcpy(&f, &v, 16);
    return f;
}

// UNAL: the reduction channel count is not a multiple of 4 (the data gradient of LSHead's 27-channel offset / mask
// convolutions): a pixel's slab is then neither 16-byte aligned nor a whole number of float4, so the pixel operand is
// fetched with four 4-byte loads per lane, each with its own channel guard.
// FINE: the staging slices are interleaved with the MFMAs of their part instruction by instruction (the slice commits
// lose their branch -- the last iteration then writes zeros into the buffer nobody reads again -- and
// sched_group_barrier groups ask for M vvv M vvv ...).  With the branch a slice is two basic blocks, hipcc interleaves
// inside one only, and the MFMAs of a slice issue back to back with its ~38 staging instructions behind them.
// Measured over the step's layer shapes (tools/ubench/conv_step, profiles/r4_conv_tiles.txt): forward 4.39 -> 4.23 ms,
// data gradient 3.29 -> 3.12 ms for the 64 x 256 tile, the same sign for 64 x 128.
// TRANS: the MFMA operands swapped back (D rows = pixels, columns = output channels): a lane owns ONE channel and sixteen
// pixels, and every store instruction writes two whole 128-byte lines (32 consecutive channels of one pixel per half-wave).
// For the backward-data GEMM of the deformable family (conv_mm_rows: N = K C = 2304 columns, plain stores, no epilogue
// terms), whose lane-per-pixel stores -- 32-byte pieces of 64 different rows per instruction -- made the L2 fetch every
// partially written line: 0.68 GB fetched by a kernel that reads 27 MB (profiles/r3_pmc_hbm.txt).
// SK: the launch has stream-K pieces (ConvArgs; conv.hip sk_plan).  A template parameter because the segment loop costs the
// plain form ~30 spilled scalar registers and 15 - 25 % on the shortest launches (profiles/r5_sk_sc1.txt).
template <int TM, int TN, int WM, int WN, int NP, bool UNAL = false, bool FINE = false, bool TRANS = false, bool SK = false>
__global__ __launch_bounds__(256, 2) void conv_mm_kernel(const ConvArgs a)
{
    using SC = SplitCfg<NP>;
    constexpr int NPL = SC::NPL;
    static_assert(WM * WN == 4, "four waves");
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = 32;
    constexpr int NLD = BM / 32;         // float4 loads per thread and chunk (8 lanes per pixel, 32 pixels per pass)
    constexpr int PLANE = BM * 64;       // bytes of one bf16 plane of a stage: BM rows of 32 bf16
    constexpr int BUF = NPL * PLANE;
    constexpr int OOB = 0x7ffffff0;      // beyond num_records: the buffer load returns 0
    extern __shared__ __align__(16) unsigned char smem[];   // 2 x BUF

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int K = a.kh * a.kw;
    const int ncc = cv_ncc(a.C), NT = cv_nt(a.Co);
    const int Tall = K * ncc;
    // ---- this workgroup's share: one whole tile, or a stream-K piece (ConvArgs) ----
    const bool is_sk = SK && (int)blockIdx.x >= a.n_dp;
    // (32-bit unit arithmetic: the host checks U * sk_n < 2^31 -- 64-bit divisions cost dozens of scalar registers here)
    int sk_u = 0, sk_end = 0, sk_start = 0, sk_U = 0;
    int sk_s = 0;
    if (is_sk) {
        sk_s = xcd_remap((int)blockIdx.x - a.n_dp, a.sk_n);
        sk_U = a.sk_tiles * Tall;
        sk_start = sk_u = (int)((unsigned)sk_U * (unsigned)sk_s / (unsigned)a.sk_n);
        sk_end = (int)((unsigned)sk_U * (unsigned)(sk_s + 1) / (unsigned)a.sk_n);
    }
    __shared__ unsigned sk_ticket;
  for (;;) {   // one pass per segment (plain tiles: exactly one)
    int work, t_begin, T;
    bool sk_partial = false;
    if (!is_sk) {
        // XCD-ordered (pixel tile, column block) with the column blocks of a pixel tile adjacent (same input rows)
        work = xcd_remap((int)blockIdx.x, a.n_dp);
        // (32-bit: Tall <= 64 taps x 64 slabs and at most 16 splits.  As a 64-bit quotient -- two scalar software divisions of
        // ~300 dependent instructions -- this was 60 % of the kernel's prologue, run by every workgroup of every launch)
        if (gridDim.z == 1) {
            t_begin = 0, T = Tall;
        } else {
            t_begin = (int)((unsigned)Tall * blockIdx.z / gridDim.z);
            T = (int)((unsigned)Tall * (blockIdx.z + 1) / gridDim.z) - t_begin;
        }
    } else {
        const int kt = (int)((unsigned)sk_u / (unsigned)Tall);
        work = a.n_dp + kt;
        t_begin = sk_u - kt * Tall;
        const int te = sk_end - kt * Tall;
        T = (te < Tall ? te : Tall) - t_begin;
        sk_partial = T != Tall;
    }
    const int ptile = work / a.colblocks + a.tile_base;
    int li = 0;
    while (li + 1 < a.nlv && ptile >= a.lv[li + 1].tile0) ++li;
    const ConvLvl &L = a.lv[li];
    const int tile_p = (ptile - L.tile0) * BM;
    const int co_blk = (work - (ptile - a.tile_base) * a.colblocks) * BN;

    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(L.x), 0, L.B * L.H * L.W * a.xpitch * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(a.wf), 0, a.wf_bytes, 0x00020000);

    // ---- pixel operand: thread = (float4 slot c4 of the 32-channel slab, pixel row prow of a 32-row pass) ----
    const int c4 = tid & 7, prow = tid >> 3;
    int pbase[NLD];                 // byte offset of the pass's pixel at tap (0, 0), channel 4 c4 (may be negative)
    unsigned long long vmask[NLD];  // one validity bit per tap (kh * kw <= 64: checked by the host)
#pragma unroll
    for (int ps = 0; ps < NLD; ++ps) {
        const int p = tile_p + ps * 32 + prow;
        const bool ok = p < L.P;
        const int HWo = L.Ho * L.Wo;
        const int b = ok ? p / HWo : 0, rem = ok ? p - b * HWo : 0;
        const int ho = rem / L.Wo, wo = rem - ho * L.Wo;
        const int iy0 = ho * a.stride - a.pad_h, ix0 = wo * a.stride - a.pad_w;
        pbase[ps] = ((b * L.H * L.W + iy0 * L.W + ix0) * a.xpitch + 4 * c4) * 4;
        unsigned long long m = 0;
        if (ok)
            for (int i = 0; i < a.kh; ++i)
                for (int j = 0; j < a.kw; ++j) {
                    const int y = iy0 + i * a.dil, x = ix0 + j * a.dil;
                    if ((unsigned)y < (unsigned)L.H && (unsigned)x < (unsigned)L.W) m |= 1ull << (i * a.kw + j);
                }
        vmask[ps] = m;
    }
    // LDS byte offset of this thread's 8 bytes in row (ps * 32 + prow): 16-byte slot (c4 >> 1) ^ ((row >> 2) & 3)
    const int st_off = prow * 64 + ((((c4 >> 1) ^ ((prow >> 2) & 3)) << 4) | ((c4 & 1) << 3));

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
    float4 xv[NLD];
    int sx_c0 = 0;
    int sx_tap = 0, sx_toff = 0;   // chunk under issue (scalars shared by its load slices)
    bool sx_cok = false;
    auto open_x = [&](const Ck &c, bool live) {   // !live: past the last chunk, no memory access
        sx_tap = c.i * a.kw + c.j;
        sx_toff = ((c.i * a.dil * L.W + c.j * a.dil) * a.xpitch + c.cc * BK) * 4;
        sx_cok = live && c.cc * BK + 4 * c4 < a.C;
        sx_c0 = c.cc * BK + 4 * c4;
    };
    auto issue_slice = [&](int ps) {
        const bool ok = ((vmask[ps] >> sx_tap) & 1ull) != 0 && sx_cok;
        if constexpr (!UNAL) {
            xv[ps] = cv_load4(xrs, ok ? pbase[ps] + sx_toff : OOB, 0);
        } else {
            const int c0 = sx_c0, vo = pbase[ps] + sx_toff;
            xv[ps].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, ok ? vo : OOB, 0, 0));
            xv[ps].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, ok && c0 + 1 < a.C ? vo + 4 : OOB, 0, 0));
            xv[ps].z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, ok && c0 + 2 < a.C ? vo + 8 : OOB, 0, 0));
            xv[ps].w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, ok && c0 + 3 < a.C ? vo + 12 : OOB, 0, 0));
        }
    };
    auto commit_slice = [&](int ps, unsigned char *buf) {
        unsigned p0[NPL], p1[NPL];
        // (Round-4 ablation, profiles/r4_conv_ablation.txt: with this split REMOVED the step's forward convolutions take
        // 4321 us against 4266 us -- the VALU work sits entirely in the shadow of the MFMAs; operands arriving pre-split would
        // buy nothing.  Removing the weight-fragment loads buys 9 %, the pixel loads 5.5 %, the barrier 2.5 %.)
        split_planes<NPL>(xv[ps].x, xv[ps].y, p0);
        split_planes<NPL>(xv[ps].z, xv[ps].w, p1);
        unsigned char *p = buf + ps * 32 * 64 + st_off;
#pragma unroll
        for (int q = 0; q < NPL; ++q) *reinterpret_cast<uint2 *>(p + q * PLANE) = make_uint2(p0[q], p1[q]);
    };
    auto issue_x = [&](const Ck &c, bool live) {
        open_x(c, live);
#pragma unroll
        for (int ps = 0; ps < NLD; ++ps) issue_slice(ps);
    };
    auto commit_x = [&](unsigned char *buf) {
#pragma unroll
        for (int ps = 0; ps < NLD; ++ps) commit_slice(ps, buf);
    };

    // ---- weight operand: fragments straight from L2.  Byte offset of (chunk t, tile nt, k-step ks, plane q):
    // ((((t * NT + nt) * 2 + ks) * NPL + q) * 1024 + lane * 16 ----
    // The wave's part (lane, wn) goes into the VGPR offset, the chunk's into a scalar one: no waterfall loop around a
    // wave-uniform-but-not-provably-so soffset.
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

    // ---- prologue: chunk 0 -> LDS buffer 0, raw pixels of chunk 1 and the weight fragments of chunk 0 in flight ----
    Ck c1;
    {
        const int k0 = t_begin / ncc;
        c1.cc = t_begin - k0 * ncc;
        c1.i = k0 / a.kw;
        c1.j = k0 - c1.i * a.kw;
    }
    // (issue order = consumption order of the steady state: pixels of t + 1, then the two weight k-steps of t, so that
    // the counted vmcnt waits of the loop leave the younger loads in flight)
    issue_x(c1, true);
    commit_x(smem);
    if (T > 1) next(c1);
    issue_x(c1, T > 1);
    issue_w(0, 0);
    issue_w(0, 1);
    __syncthreads();

    // fragment read address: row = wm * TM * 32 + i * 32 + (lane & 31), slot = (ks * 2 + (lane >> 5)) ^ ((row >> 2) & 3)
    const int frow = wm * TM * 32 + (lane & 31);
    const int fsw = (frow >> 2) & 3;   // (row >> 2) & 3 is the same for every i (rows differ by 32)

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
                    acc[j][i] = TRANS ? mfma_bf16(Xf[ks][i][SC::pa(prod)], Wf[ks][j][SC::pb(prod)], acc[j][i])
                                      : mfma_bf16(Wf[ks][j][SC::pb(prod)], Xf[ks][i][SC::pa(prod)], acc[j][i]);
    };

    // One iteration = one chunk.  The fences pin the order of its phases; inside a phase hipcc schedules freely.  Left
    // alone it sinks all sixteen loads to the end of the iteration (the next one then opens with a full memory-latency
    // wait) and reads every fragment right before its first MFMA.
    //   A  fragment reads of k-step 0 (their LDS latency hides behind B)
    //   B  split the raw pixels of chunk t + 1 -> the other LDS buffer; issue the pixel loads of chunk t + 2
    //   C  fragment reads of k-step 1, MFMAs of k-step 0, then the weight loads of (t + 1, 0) into the dead registers
    //   D  MFMAs of k-step 1, weight loads of (t + 1, 1); barrier
    // Loads are issued in the order they are consumed, so every counted vmcnt wait leaves the younger ones in flight:
    // each load has at least half an iteration (24 MFMAs of this wave plus the partner workgroup's share of the SIMD).
#ifdef LSNET_OLD_LOOP
    for (int t = 0; t < T; ++t) {
        const unsigned char *bc = smem + (t & 1) * BUF;
        unsigned char *bn = smem + ((t & 1) ^ 1) * BUF;
        read_x(bc, 0);
        read_x(bc, 1);
        if (t + 2 < T) next(c1);
        open_x(c1, t + 2 < T);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int NM = NP * TN * TM;
#pragma unroll
        for (int ps = 0; ps < NLD; ++ps) {
#pragma unroll
            for (int m = ps * NM / NLD; m < (ps + 1) * NM / NLD; ++m) {
                const int prod = m / (TN * TM), j = (m / TM) % TN, i = m % TM;
                acc[j][i] = TRANS ? mfma_bf16(Xf[0][i][SC::pa(prod)], Wf[0][j][SC::pb(prod)], acc[j][i])
                                  : mfma_bf16(Wf[0][j][SC::pb(prod)], Xf[0][i][SC::pa(prod)], acc[j][i]);
            }
            if (FINE || t + 1 < T) commit_slice(ps, bn);
            issue_slice(ps);
            if constexpr (FINE) {
#pragma unroll
                for (int g = 0; g < NM / NLD; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x006, 3, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        issue_w(t + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_block(1);
        issue_w(t + 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
#else
    // Round 5: the barrier sits in the MIDDLE of the iteration and the k-step-0 fragments of chunk t + 1 are read right
    // behind it, under the MFMAs of k-step 1 (their registers are dead by then); the k-step-1 fragments of chunk t are
    // read at the top, under the MFMAs of k-step 0.  No MFMA waits for an LDS read issued just in front of it any more.
    //   top   fragment reads of (t, k-step 1)
    //   B     MFMAs of k-step 0 in NLD parts; behind each: split 32 pixels of chunk t + 1 -> the other LDS buffer, fetch
    //         the same rows of chunk t + 2; then the weight loads of (t + 1, 0)
    //   ----  barrier: buffer t + 1 is complete; nobody reads buffer t any more (its k-step-0 fragments were taken in
    //         iteration t - 1, its k-step-1 fragments at the top of this one and the barrier waits for them)
    //   D     fragment reads of (t + 1, k-step 0), MFMAs of k-step 1, weight loads of (t + 1, 1)
    read_x(smem, 0);   // (chunk 0: the prologue's barrier has passed)
    for (int t = 0; t < T; ++t) {
        const unsigned char *bc = smem + (t & 1) * BUF;
        unsigned char *bn = smem + ((t & 1) ^ 1) * BUF;
        read_x(bc, 1);
        if (t + 2 < T) next(c1);
        open_x(c1, t + 2 < T);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int NM = NP * TN * TM;
#pragma unroll
        for (int ps = 0; ps < NLD; ++ps) {
#pragma unroll
            for (int m = ps * NM / NLD; m < (ps + 1) * NM / NLD; ++m) {
                const int prod = m / (TN * TM), j = (m / TM) % TN, i = m % TM;
                acc[j][i] = TRANS ? mfma_bf16(Xf[0][i][SC::pa(prod)], Wf[0][j][SC::pb(prod)], acc[j][i])
                                  : mfma_bf16(Wf[0][j][SC::pb(prod)], Xf[0][i][SC::pa(prod)], acc[j][i]);
            }
            if (FINE || t + 1 < T) commit_slice(ps, bn);   // registers hold the raw pixels of chunk t + 1
            issue_slice(ps);
            if constexpr (FINE) {
#pragma unroll
                for (int g = 0; g < NM / NLD; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x006, 3, 0);   // three vector / scalar ALU instructions
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        issue_w(t + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        read_x(bn, 0);   // (past the last chunk: whatever the buffer holds, never used)
        mfma_block(1);
        issue_w(t + 1, 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();   // (the next segment's prologue writes buffer 0: every wave is past its last fragment read)
#endif

    // ---- stream-K: a piece that holds only part of the tile's sum ----
    // Visibility across the eight L2s without flushing them: the slot is written and read with AGENT-scope accesses (sc1:
    // written through to / fetched from the coherence point), the writer waits for its stores (vmcnt) before the ticket is
    // drawn.  A release / acquire FENCE pair instead (__threadfence: buffer_wbl2 + buffer_inv of the whole L2, from 512
    // workgroups) cost 150 - 250 us per launch (profiles/r5_sk_fence.txt).
    bool do_out = true;
    if (SK && sk_partial) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        constexpr int SLOT = BM * BN;   // floats; element (q, tid) of a slot = float4 number q of thread tid (coalesced)
        constexpr int SC1 = 16;         // cache-policy bit 4 of the raw buffer intrinsics on gfx940+: agent scope
        const int kt = work - a.n_dp;
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(a.sk_part, 0, 2 * a.sk_n * SLOT * 4, 0x00020000);
        {
            const int soff = (2 * sk_s + (sk_u == sk_start ? 0 : 1)) * (SLOT * 4);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = {acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), prs,
                                                               (((j * TM + i) * 4 + g) * 256 + tid) * 16, soff, SC1);
                    }
        }
        // contributors of tile kt: the pieces that own its first and its last chunk, and those between
        const int u_lo = kt * Tall, u_hi = u_lo + Tall - 1;
        const int s_lo = (int)(((unsigned)(u_lo + 1) * (unsigned)a.sk_n + (unsigned)sk_U - 1u) / (unsigned)sk_U) - 1;
        const int s_hi = (int)(((unsigned)(u_hi + 1) * (unsigned)a.sk_n + (unsigned)sk_U - 1u) / (unsigned)sk_U) - 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's slot elements have reached the coherence point
        __syncthreads();
        if (tid == 0) sk_ticket = __hip_atomic_fetch_add(a.sk_cnt + kt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        do_out = (int)sk_ticket == s_hi - s_lo;
        if (do_out) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
            for (int c = s_lo; c <= s_hi; ++c) {   // chunk order = piece order
                const int cs = (int)((unsigned)sk_U * (unsigned)c / (unsigned)a.sk_n);
                const int soff = (2 * c + (cs >= u_lo ? 0 : 1)) * (SLOT * 4);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x4 pv[TM * 4];   // a column of tiles in flight, then the additions
#pragma unroll
                    for (int q = 0; q < TM * 4; ++q)
                        pv[q] = __builtin_bit_cast(
                            f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, ((j * TM * 4 + q) * 256 + tid) * 16, soff, SC1));
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[j][i][4 * g + e] += pv[i * 4 + g][e];
                }
            }
            if (tid == 0) __hip_atomic_store(a.sk_cnt + kt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed
        }
    }

    // ---- epilogue ----
    const bool partial = a.ksplit > 1;
    const bool vec_ok = (a.Co & 3) == 0;
    if (do_out) {
    if constexpr (TRANS) {
        // lane = output channel (lane & 31) of tile j, pixels mfma32_row(r, lane) of tile i: plain stores, whole lines.
        // Round 6: through ONE buffer descriptor of the level's rows -- a pixel past the level's end lies past the descriptor
        // and its store is dropped, so no branch and no pointer arithmetic sits between the 64 stores of a thread.  (Written
        // with pointers the epilogue was 930 instructions for 64 stores: 96 exec branches and, per row, a scalar RELOAD of the
        // level's output pointer with its wait -- ~2 us per workgroup behind a main loop of 5.)
        // (nontemporal, cache-policy bit 1 = nt: the 413 MB of column gradients of a tower launch stream THROUGH the L2s that
        // hold the weight image and the pixel rows every column block of a tile re-reads; profiles/r6_gemm_l2.txt)
        const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(L.out, 0, L.P * a.Co * 4, 0x00020000);
        const int cbase = (co_blk + wn * TN * 32 + (lane & 31)) * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pix = tile_p + wm * TM * 32 + i * 32 + mfma32_row(r, lane);
                const int voff = pix < L.P ? pix * a.Co * 4 + cbase : OOB;   // (a whole row past the end: past the descriptor)
                // (conv.hip conv_mm_rows: every column of a TRANS launch exists -- Co is a multiple of the 256-column tile)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float v = acc[j][i][r];   // (a copy: __builtin_bit_cast of the vector ELEMENT read element 0 for every r)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ors, voff + j * 128, 0, 2);
                }
            }
    } else {
    // lane = pixel (lane & 31) of tile i, output channels 8 g + 4 (lane >> 5) + (0..3) of tile j
    // Round 6: the epilogue's operands (bias, residual, gate) are FETCHED FIRST, all quads of a pixel tile at once, through
    // buffer loads whose masked-out lanes point past the tensor -- no branch between the loads, one wait per operand kind.
    // (Written quad by quad -- load bias, wait, load residual, wait, load gate, wait, store -- every one of the sixteen quads
    // of a thread paid up to three memory round trips in a row: 61 vs 38 us for the 1024 -> 256 data gradient of stage 3
    // with and without its residual and gate, profiles/r6_instep_vs_isolated.txt "warm" vs "warm-epi".)
    const size_t out_px = a.ostep ? (size_t)L.B * a.OH * a.OW : (size_t)L.P;
    const int ep_bytes = (int)(out_px * a.Co * 4);   // (conv.hip conv_check: the output fits 32-bit byte offsets)
    const bool fin = !partial;   // bias, residual, ReLU and gate belong to the finished sum
    const bool has_bias = fin && a.bias != nullptr, has_res = fin && L.res != nullptr, has_gate = fin && L.gate != nullptr;
    // the stores go through a buffer descriptor as well (the output, or this split's partial tiles): a pixel past the level's
    // end or a channel past Co becomes an offset past the descriptor and the store is dropped -- no branch per quad
    const __amdgpu_buffer_rsrc_t ors =
        partial ? __builtin_amdgcn_make_buffer_rsrc(a.part + (size_t)blockIdx.z * a.part_rows * a.Co, 0, a.part_rows * a.Co * 4, 0x00020000)
                : __builtin_amdgcn_make_buffer_rsrc(L.out, 0, ep_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int pix = tile_p + wm * TM * 32 + i * 32 + (lane & 31);
        if (pix >= L.P) continue;
        size_t opix = pix;
        if (a.ostep) {
            const int HWo = L.Ho * L.Wo;
            const int b = pix / HWo, rem = pix - b * HWo;
            const int ho = rem / L.Wo, wo = rem - ho * L.Wo;
            opix = ((size_t)b * a.OH + a.oy0 + ho * a.ostep) * a.OW + a.ox0 + wo * a.ostep;
        }
        const int orow_off = (partial ? (pix - a.part_pix0) * a.Co : (int)opix * a.Co) * 4;
        // one finished quad -> memory: 16 bytes, or element by element when a pixel's row is not a whole number of quads
        auto put = [&](int co, const float (&v)[4]) __attribute__((always_inline)) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            if (vec_ok) {
                const u32x4 q = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                __builtin_amdgcn_raw_buffer_store_b128(q, ors, co < a.Co ? orow_off + co * 4 : OOB, 0, 0);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[e]), ors, co + e < a.Co ? orow_off + (co + e) * 4 : OOB, 0, 0);
            }
        };
        if (!has_bias && !has_res && !has_gate) {   // nothing to fetch (partial tiles, plain data gradients)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = co_blk + (wn * TN + j) * 32 + 8 * g + 4 * (lane >> 5);
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (fin && a.relu) ? fmaxf(acc[j][i][4 * g + e], 0.f) : acc[j][i][4 * g + e];
                    put(co, v);
                }
        } else {
            float4 bv[TN * 4], rv[TN * 4], gv[TN * 4];
            const int obase = (int)(opix * a.Co) * 4;
            // (a quad beyond Co -- the padded column tiles of a narrow output -- reads past the bias, i.e. zero, and the first
            // channels of the NEXT pixel of the residual / gate, i.e. memory of the same tensor or, behind the last pixel,
            // zero: it is never stored, so no select sits between the address and the load.  The same holds for the last
            // elements of a quad that straddles Co (Co % 4 != 0: the 27 offset / mask channels of LSHead).)
            const int cobase = (co_blk + wn * TN * 32 + 4 * (lane >> 5)) * 4;
            // 16-byte loads where rows are whole quads, four 4-byte loads per quad otherwise (a row of 27 floats starts anywhere)
            auto fetch = [&](__amdgpu_buffer_rsrc_t rs, int base, float4 (&dst)[TN * 4]) __attribute__((always_inline)) {
                if (vec_ok) {
#pragma unroll
                    for (int q = 0; q < TN * 4; ++q) dst[q] = cv_load4(rs, base + ((q >> 2) * 32 + (q & 3) * 8) * 4, 0);
                } else {
#pragma unroll
                    for (int q = 0; q < TN * 4; ++q) {
                        const int o = base + ((q >> 2) * 32 + (q & 3) * 8) * 4;
                        dst[q].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, o, 0, 0));
                        dst[q].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, o + 4, 0, 0));
                        dst[q].z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, o + 8, 0, 0));
                        dst[q].w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, o + 12, 0, 0));
                    }
                }
            };
            // (zero first, although every use sits behind its operand's flag: left unset hipcc re-serialises the loads -- the
            // step's forward shapes 3.73 -> 4.03 ms in tools/ubench/conv_step)
#pragma unroll
            for (int q = 0; q < TN * 4; ++q) bv[q] = rv[q] = gv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            // (each descriptor lives only around its own loads: twelve scalar registers held across the whole epilogue cost
            // the stream-K instantiations ~100 more spilled ones)
            if (has_bias)
                fetch(__builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.bias), 0, a.Co * 4, 0x00020000), cobase, bv);
            if (has_res)
                fetch(__builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(L.res), 0, ep_bytes, 0x00020000), obase + cobase, rv);
            if (has_gate)
                fetch(__builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(L.gate), 0, ep_bytes, 0x00020000), obase + cobase, gv);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = co_blk + (wn * TN + j) * 32 + 8 * g + 4 * (lane >> 5);
                    const float4 b4 = bv[j * 4 + g], r4 = rv[j * 4 + g], g4 = gv[j * 4 + g];
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[j][i][4 * g + e];
                    if (has_bias) v[0] += b4.x, v[1] += b4.y, v[2] += b4.z, v[3] += b4.w;
                    if (has_res) v[0] += r4.x, v[1] += r4.y, v[2] += r4.z, v[3] += r4.w;
                    if (fin && a.relu)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    if (has_gate) {
                        v[0] = g4.x > 0.f ? v[0] : 0.f, v[1] = g4.y > 0.f ? v[1] : 0.f;
                        v[2] = g4.z > 0.f ? v[2] : 0.f, v[3] = g4.w > 0.f ? v[3] : 0.f;
                    }
                    put(co, v);
                }
        }
    }
    }
    }
    if (!SK || !is_sk) break;
    sk_u += T;
    if (sk_u >= sk_end) break;
  }
}

// out[e] = sum_z part[z * n + e] + bias[e % Co] (ReLU): the second pass of a split reduction; n = P * Co
static __global__ void conv_splitk_reduce_kernel(const float *__restrict__ part, float *out, const float *__restrict__ bias,
                                                 const float *res, const float *__restrict__ gate, int n, int Co, int ks,
                                                 int relu)
{
    const bool v4 = (n & 3) == 0 && (Co & 3) == 0;
    if (v4) {
        for (int e = (blockIdx.x * blockDim.x + threadIdx.x) * 4; e < n; e += gridDim.x * blockDim.x * 4) {
            float4 s = *reinterpret_cast<const float4 *>(part + e);
#pragma unroll 4
            for (int z = 1; z < ks; ++z) {
                const float4 p = *reinterpret_cast<const float4 *>(part + (size_t)z * n + e);
                s.x += p.x, s.y += p.y, s.z += p.z, s.w += p.w;
            }
            if (bias) {
                const float4 b = *reinterpret_cast<const float4 *>(bias + e % Co);
                s.x += b.x, s.y += b.y, s.z += b.z, s.w += b.w;
            }
            if (res) {
                const float4 r = *reinterpret_cast<const float4 *>(res + e);
                s.x += r.x, s.y += r.y, s.z += r.z, s.w += r.w;
            }
            if (relu) s.x = fmaxf(s.x, 0.f), s.y = fmaxf(s.y, 0.f), s.z = fmaxf(s.z, 0.f), s.w = fmaxf(s.w, 0.f);
            if (gate) {
                const float4 g = *reinterpret_cast<const float4 *>(gate + e);
                s.x = g.x > 0.f ? s.x : 0.f, s.y = g.y > 0.f ? s.y : 0.f, s.z = g.z > 0.f ? s.z : 0.f, s.w = g.w > 0.f ? s.w : 0.f;
            }
            *reinterpret_cast<float4 *>(out + e) = s;
        }
    } else {
        for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
            float s = part[e];
            for (int z = 1; z < ks; ++z) s += part[(size_t)z * n + e];
            if (bias) s += bias[e % Co];
            if (res) s += res[e];
            if (relu) s = fmaxf(s, 0.f);
            if (gate && !(gate[e] > 0.f)) s = 0.f;
            out[e] = s;
        }
    }
}

}  // namespace lsn
