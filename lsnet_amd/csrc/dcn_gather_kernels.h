// Atomic-free backward-data of the deformable family (included by dcn_kernels.h, which holds DcnArgs / Tap / Lvl): anchor
// lists (bin / scan / fill / sort), per-anchor corner sums, the four-anchor combine, offset / mask gradients from the
// corner dot products, and the fmaf-chain column gradients of grouped calls.  This is the DEFAULT path of every math mode;
// what is left in dcn_kernels.h are the general-shape kernels (channel counts the kernels of dcn_mm_kernels.h do not
// take, callers without the gather workspace) and the exact-mode GEMMs.
#pragma once

namespace lsn {

// =============================================================================================
// grad_input without atomics: anchor lists + gather (the COLBUF path of dcn_bwd_data_xn_kernel).
//
// A sample (pixel, tap[, deformable group]) at position (py, px) touches the four input pixels around its
// ANCHOR (y0, x0) = floor(py, px), y0 in [-1, H-1], x0 in [-1, W-1].  Input pixel q = (y, x) therefore receives
//   from anchor (y, x)     the (1-ly)(1-lx) corner,   from (y, x-1)   the (1-ly) lx corner,
//   from anchor (y-1, x)   the ly (1-lx) corner,      from (y-1, x-1) the ly lx corner
// of every sample anchored there.  dcn_bin_kernel counts the samples of every anchor (integer atomics, 1 per sample
// against 4 x C float atomics per sample before) and records each sample's rank; a single-block scan turns the counts
// into list offsets; dcn_fill_kernel writes {sample, ly, lx} entries; dcn_sort_lists_kernel orders each list by sample
// id (so that the summation order is fixed); dcn_gather_kernel walks the four lists of every input pixel, reads the
// mask-weighted column-gradient rows (C contiguous floats each) and writes grad_input once.
// Levels that scatter into the same grad_input buffer (the pyramid op: one source map for several destination
// levels) share one anchor grid, so their contributions are summed here instead of by separate tensor adds.
// =============================================================================================
struct __align__(16) GEntry {
    int s;        // sample id = (prow0 + pix) * KD + dgi * K + k
    int pad;
    float ly, lx;
};

struct GatherGrp {
    float *gx;
    const float *x;   // the input map grad_input belongs to (corner sums of the offset / mask gradients)
    int B, H, W;
    int abase;   // first anchor id: anchors (B, H+1, W+1), (y0+1, x0+1) row-major
    int blk0;    // first 4x4 pixel block of this group in the launch-wide numbering
};
struct GatherArgs {
    GatherGrp g[MAXLV];
    int ng, NB;         // groups, total 4x4 pixel blocks
    int C, K, KD, dg;
    const float *gcol;
    const int *start;   // [anchors + 1]
    const GEntry *ent;
    // raw != 0: gcol holds the column gradients WITHOUT the modulation scalar (the GEMM of conv_kernels.h wrote them); an
    // entry's scalar (GEntry::pad) is applied here, and Hb (if not NULL) receives, per sample and corner, the dot product of
    // the sample's column-gradient row with that corner's input pixel: [sample][corner 00, 01, 10, 11]
    int raw;
    float *Hb;
};

// Groups whose lists are LONG (the pyramid op: source level P5 receives the samples of target levels P3 .. P7, 2600 per
// 4x4 pixel block) are gathered per ANCHOR instead of per pixel block: a map of 25 x 42 pixels has 154 blocks but 2236
// anchors -- 14 x the workgroups walking lists that are as many times shorter.
//   dcn_anchor_sum_kernel     one workgroup per anchor: S[anchor][dy][dx][C] = sum over the anchor's entries of the
//                             bilinear corner weight x the entry's column-gradient row (four waves share the list and
//                             meet in LDS in a fixed order);
//   dcn_anchor_combine_kernel grad_input[y][x] = S[y-1][x-1][1][1] + S[y-1][x][1][0] + S[y][x-1][0][1] + S[y][x][0][0]
//                             (anchor coordinates shifted by one as in the lists): each element written once.
struct AnchorGrp {
    float *gx;
    const float *x;
    int B, H, W;
    int abase;   // first anchor id in the launch-wide numbering (lists)
    int a0;      // first anchor of this group in the S buffer
};
struct AnchorArgs {
    AnchorGrp g[MAXLV];
    int ng, NA;         // groups, anchors of all of them
    int C, K, KD, dg;
    const float *gcol;
    const int *start;
    const GEntry *ent;
    float *S;           // [NA][4][C]
    int raw;            // as in GatherArgs
    float *Hb;
    // ncb > 1 (round 6; C > 256 with every group on this path: the 512 .. 2048-channel deformable layers of config 4's
    // backbone): the 256-channel blocks of an anchor are blockIdx.y -- a map of 50 x 84 pixels has 8.6 k anchors, two waves per
    // SIMD, each walking its list once per block in sequence.  Block y leaves its share of the corner dot products in slot y of
    // Hb ([ncb][samples][4], hb_slot floats apart); dcn_offgrad_kernel adds the slots in block order.
    int ncb;
    long long hb_slot;
};

// Sixteen wave-wide sums at once: in, per lane, d[0..15]; out, in lane i (every 16-lane row alike), the sum over the 64
// lanes of d[i & 15].  Transposing butterfly: at the step of lane bit t a lane keeps the half of its values whose index has
// bit t equal to its own lane bit and hands the other half to the lane that keeps those (quad permutes for bits 0 / 1,
// row rotations by 4 / 8 lanes for bits 2 / 3: a rotation is a bijection between the two classes, which is all a sum
// needs), so the value count halves per step -- 15 adds and 30 selects instead of 16 x 6 dependent adds.  The four rows
// are then added through two lane swaps.  Fixed order of additions: deterministic.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave64_sum16(const float (&d)[16], int lane)
{
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
    float w[8], w2[4], z[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = (b0 ? d[2 * i + 1] : d[2 * i]) + dpp_f32<0xB1>(b0 ? d[2 * i] : d[2 * i + 1]);     // quad_perm [1,0,3,2]
#pragma unroll
    for (int i = 0; i < 4; ++i) w2[i] = (b1 ? w[2 * i + 1] : w[2 * i]) + dpp_f32<0x4E>(b1 ? w[2 * i] : w[2 * i + 1]);   // quad_perm [2,3,0,1]
#pragma unroll
    for (int i = 0; i < 2; ++i) z[i] = (b2 ? w2[2 * i + 1] : w2[2 * i]) + dpp_f32<0x124>(b2 ? w2[2 * i] : w2[2 * i + 1]);   // row_ror:4
    float r = (b3 ? z[1] : z[0]) + dpp_f32<0x128>(b3 ? z[0] : z[1]);                                                    // row_ror:8
    r += __shfl_xor(r, 16);
    r += __shfl_xor(r, 32);
    return r;
}

__device__ __forceinline__ const Lvl &find_level_by_row(const DcnArgs &a, int prow)
{
    int li = 0;
    while (li + 1 < a.nlv && prow >= a.lv[li + 1].prow0) ++li;
    return a.lv[li];
}

// Column gradients of a GROUPED deformable convolution (ResNeXt-101 64x4d-DCN, BASELINE config 4: 64 groups of 8 / 16 /
// 32 channels; resnext.py:11-83 + deform_conv_cuda.cpp:747-748 per group):
//   gcol[(prow, k)][c] = sum_{j < Co / groups} gout[prow][g Cog + j] * w[g Cog + j][k][c - g Cg],   g = c / Cg
// -- UNWEIGHTED, for the gather pass above (anchor lists, per-anchor corner sums, offset / mask gradients from the same
// rows), which serves grouped calls with this kernel in front instead of the dense GEMM.  Per group the product is 8 ..
// 32 wide: 1/16 .. 1 of one MFMA tile and 0.5 .. 2 MAC per gathered byte, so exact fp32 fmaf chains on the vector ALUs
// (reference arithmetic, every math mode); the launch is bound by writing gcol.  Replaces round 1's atomic scatter for
// these calls: grad_input of config 4 is bit-reproducible.  A thread = four consecutive channels of one (pixel, tap) row.
__global__ __launch_bounds__(256) void dcn_gcol_grouped_kernel(const DcnArgs a, long long nquads)
{
    const int K = a.kh * a.kw, C = a.C, Cg = C / a.groups, Cog = a.Co / a.groups, C4 = C >> 2;
    for (long long id = blockIdx.x * (long long)blockDim.x + threadIdx.x; id < nquads; id += (long long)gridDim.x * blockDim.x) {
        const long long row = id / C4;                  // (launch-wide pixel row, tap)
        const int c = (int)(id - row * C4) * 4;
        const int prow = (int)(row / K), k = (int)(row - (long long)prow * K);
        const Lvl &L = find_level_by_row(a, prow);
        const int g = c / Cg, ci = c - g * Cg;
        const float *go = L.gout + (size_t)(prow - L.prow0) * a.Co + (size_t)g * Cog;
        const float *w = a.w + ((size_t)g * Cog * K + k) * Cg + ci;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < Cog; ++j) {
            const float gv = go[j];
            const float4 wv = *reinterpret_cast<const float4 *>(w + (size_t)j * K * Cg);
            s.x = fmaf(gv, wv.x, s.x), s.y = fmaf(gv, wv.y, s.y), s.z = fmaf(gv, wv.z, s.z), s.w = fmaf(gv, wv.w, s.w);
        }
        *reinterpret_cast<float4 *>(a.gcol + (size_t)row * C + c) = s;
    }
}

// one thread per sample: anchor, rank inside the anchor's list, fractions
// (Walking the samples tap-major like the table -- coalesced 32-byte stores, the integer atomics of a wave spread over 64
// neighbouring anchors -- measured no difference in the round-4 A/B: 604.9 vs 607.0 us per tower backward.)
__global__ void dcn_bin_kernel(const DcnArgs a, int nsamples, int *__restrict__ cnt, int *__restrict__ sanchor,
                               int *__restrict__ srank, float2 *__restrict__ sfrac, Tap *__restrict__ gtap)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;   // sample id (prow * KD + kd)
    if (s >= nsamples) return;
    const int K = a.kh * a.kw, KD = K * a.dg;
    const int prow = s / KD, kd = s - prow * KD;
    const int dgi = kd / K, k = kd - dgi * K;
    const Lvl &L = find_level_by_row(a, prow);
    int anchor = -1, rank = 0;
    float2 fr = make_float2(0.f, 0.f);
    int4 yx = make_int4(0, 0, 0, 0);
    const Tap t = make_tap_ex(a, L, prow - L.prow0, k, dgi, &yx);
    gtap[(size_t)kd * a.gtap_rows + prow] = t;   // k-major: a 32-pixel step of one tap is 1 KB contiguous
    if (s == 0) {
        Tap z = {};
        gtap[(size_t)KD * a.gtap_rows] = z;   // the "no sample" entry behind the table
    }
    if (L.gx != nullptr && t.flags) {
        const int y0 = (t.flags & 3) ? yx.x : -1, x0 = (t.flags & 5) ? yx.y : -1;   // unclamped floor(py), floor(px)
        const int HWo = L.Ho * L.Wo;
        const int b = (prow - L.prow0) / HWo;
        anchor = L.abase + (b * (L.H + 1) + y0 + 1) * (L.W + 1) + x0 + 1;
        rank = atomicAdd(&cnt[anchor], 1);
        fr = make_float2(t.ly, t.lx);
    }
    sanchor[s] = anchor;
    srank[s] = rank;
    sfrac[s] = fr;
}

// exclusive prefix sum of cnt[0..n) into start[0..n], one workgroup of 1024 threads
__global__ __launch_bounds__(1024) void dcn_scan_kernel(const int *__restrict__ cnt, int *__restrict__ start, int n)
{
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int b = min(tid * per, n), e = min(b + per, n);
    int sum = 0;
    for (int i = b; i < e; ++i) sum += cnt[i];
    part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {   // Hillis-Steele inclusive scan
        const int v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - sum;
    for (int i = b; i < e; ++i) {
        start[i] = run;
        run += cnt[i];
    }
    if (tid == 1023) start[n] = part[1023];
}

__global__ void dcn_fill_kernel(int nsamples, const int *__restrict__ start, const int *__restrict__ sanchor,
                                const int *__restrict__ srank, const float2 *__restrict__ sfrac, GEntry *__restrict__ ent,
                                const Tap *__restrict__ gtap, int KD, int gtap_rows)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsamples) return;
    const int an = sanchor[s];
    if (an < 0) return;
    GEntry e;
    const int prow = s / KD, kd = s - prow * KD;
    e.s = s, e.pad = __float_as_int(gtap[(size_t)kd * gtap_rows + prow].m);   // the sample's modulation scalar
    const float2 f = sfrac[s];
    e.ly = f.x, e.lx = f.y;
    ent[start[an] + srank[s]] = e;
}

// one wave per anchor list: every list is rewritten in ascending sample order (the arrival order is that of the integer
// atomics of dcn_bin_kernel: different on every run).  Up to 64 entries: ranks by 64 lane reads.  Longer lists (hundreds of
// samples converging on one landmark in the pyramid op): a lane ranks its every-64th entries against the whole list, 64
// keys at a time, writes them in order to `tmp` and the wave copies the range back -- n^2 / 64 compares per lane, ~10 us
// for 700 entries.  tmp == NULL: long lists keep their arrival order (round 2's behaviour).
__global__ __launch_bounds__(256) void dcn_sort_lists_kernel(int nanchors, const int *__restrict__ start, GEntry *__restrict__ ent,
                                                             GEntry *__restrict__ tmp)
{
    const int lane = threadIdx.x & 63;
    const int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    for (int an = wid; an < nanchors; an += nw) {
        const int b = start[an], n = start[an + 1] - b;
        if (n < 2) continue;
        if (n > 64) {
            if (tmp == nullptr) continue;
            for (int i0 = 0; i0 < n; i0 += 64) {          // this lane's entry i0 + lane
                const bool mine = i0 + lane < n;
                GEntry e = {};
                if (mine) e = ent[b + i0 + lane];
                int rank = 0;
                for (int j0 = 0; j0 < n; j0 += 64) {      // against keys j0 .. j0 + 63
                    const int kj = j0 + lane < n ? ent[b + j0 + lane].s : 0x7fffffff;
                    const int m = min(64, n - j0);
                    for (int j = 0; j < m; ++j) rank += (__builtin_amdgcn_readlane(kj, j) < e.s) ? 1 : 0;
                }
                if (mine) tmp[b + rank] = e;
            }
            __threadfence();
            for (int i = lane; i < n; i += 64) ent[b + i] = tmp[b + i];
            continue;
        }
        GEntry e = {};
        if (lane < n) e = ent[b + lane];
        const int key = lane < n ? e.s : 0x7fffffff;
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += (__builtin_amdgcn_readlane(key, j) < key) ? 1 : 0;   // sample ids are distinct
        if (lane < n) ent[b + rank] = e;
    }
}

// One wave per 4x4 block of input pixels; lane = 4 consecutive channels (256 channels per pass).  The 5x5 anchors around
// the block are walked once each: an entry's column-gradient row (C contiguous floats) is loaded ONCE and added, with
// its four bilinear corner weights, to the pixels of the block it touches -- 25 lists for 16 pixels instead of the 64 a
// pixel-by-pixel gather reads (measured before: the gather fetched 3.7 x the column-gradient buffer from the fabric).
// Per anchor the four corner sums S[dy][dx] are formed in registers (their index does not depend on where the anchor
// lies), then folded into the block's 16 accumulators, which live in LDS because THEIR index does: the anchor loop stays
// rolled (unrolled 25 x for register accumulators, the kernel outgrew the instruction cache once the corner sums of the
// offset gradients joined it: 150 -> 570 us on the tower launch).
// With ga.Hb the wave also forms, per entry and corner that lies in its block, the dot product of the entry's row with
// that corner's input pixel (the H sums of grad_offset / grad_mask, dcn_offgrad_kernel): the row is in registers anyway.
// Fixed order: anchors row-major, entries by sample id.
constexpr int GT = 4;   // block edge
// NW = 1: one wave per 4x4 pixel block (four blocks per workgroup) -- short lists, e.g. the tower launch with ~140 entries
//   per block.
// NW = 4: one workgroup per pixel block.  Its four waves share the block's entries -- wave w takes every 4th group of GU
//   entries, counted along the 25 lists -- and meet in LDS: the longest chain of dependent loads of a block is a quarter
//   of what one wave walks (the pyramid launch gives source levels P4 .. P7 500-700 entries per block).  The partial sums
//   are combined in the fixed order (w0 + w2) + (w1 + w3): deterministic.
template <int NW>
__global__ __launch_bounds__(256) void dcn_gather_kernel(const GatherArgs ga)
{
    constexpr int GU = 4;
    __shared__ float4 accs[4][GT * GT][64];   // per wave: 16 pixels x 64 lanes (64 KB per workgroup)
    const int lane = threadIdx.x & 63, wave_id = threadIdx.x >> 6;
    const int wave = NW == 4 ? wave_id : 0;   // position among the waves that share a pixel block
    const int C = ga.C, K = ga.K, KD = ga.KD;
    const int cpdg = C / ga.dg;
    // NW = 4 walks the blocks LAST GROUP FIRST: in the pyramid launch the coarse source levels (P5 .. P7) have a few dozen
    // blocks with thousands of entries each -- a chain of dependent row reads hundreds of microseconds long.  Started last
    // they are the launch's tail; started first the thousands of short fine-level blocks fill in behind them.
    const int bi = NW == 4 ? ga.NB - 1 - xcd_remap(blockIdx.x, gridDim.x) : xcd_remap(blockIdx.x, gridDim.x) * 4 + wave_id;
    if (bi >= ga.NB) return;
    int gi = 0;
    while (gi + 1 < ga.ng && bi >= ga.g[gi + 1].blk0) ++gi;
    const GatherGrp &G = ga.g[gi];
    const int nbx = (G.W + GT - 1) / GT, nby = (G.H + GT - 1) / GT;
    const int lb_ = bi - G.blk0;
    const int b = lb_ / (nbx * nby), rem = lb_ - b * nbx * nby;
    const int y0 = (rem / nbx) * GT, x0 = (rem % nbx) * GT;
    float4(*acc)[64] = accs[wave_id];
    for (int cb = 0; cb < C; cb += 256) {
        const int c = cb + lane * 4;
#pragma unroll
        for (int p = 0; p < GT * GT; ++p) acc[p][lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        int grp = 0;   // running group counter of the block (wave-uniform)
        for (int an_i = 0; an_i < (GT + 1) * (GT + 1); ++an_i) {
            const int ai = an_i / (GT + 1), aj = an_i - ai * (GT + 1);
            const int ay = y0 - 1 + ai, ax = x0 - 1 + aj;   // anchor = floor of the sample position, >= -1
            if (ay > G.H - 1 || ax > G.W - 1) continue;     // (wave-uniform)
            const int an = G.abase + (b * (G.H + 1) + ay + 1) * (G.W + 1) + ax + 1;
            const int lb = __builtin_amdgcn_readfirstlane(ga.start[an]);
            const int le = __builtin_amdgcn_readfirstlane(ga.start[an + 1]);
            if (lb == le) continue;
            // the anchor's corner pixels (ay + dy, ax + dx): in this block?  (then also inside the map, except for the
            // block's own overhang past the map edge)
            bool inb[2][2];
            float4 xa[2][2], S[2][2];
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int pi = ai - 1 + dy, pj = aj - 1 + dx;
                    inb[dy][dx] = pi >= 0 && pi < GT && pj >= 0 && pj < GT && y0 + pi < G.H && x0 + pj < G.W;
                    S[dy][dx] = make_float4(0.f, 0.f, 0.f, 0.f);
                    xa[dy][dx] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ga.Hb != nullptr && inb[dy][dx] && c < C)
                        xa[dy][dx] = *reinterpret_cast<const float4 *>(G.x + ((size_t)(b * G.H + ay + dy) * G.W + ax + dx) * C + c);
                }
            for (int base = lb; base < le; base += 64) {
                const int n = min(64, le - base);
                const int ngrp = (n + GU - 1) / GU;
                // groups of this batch that are this wave's: first one at offset (wave - grp) mod NW
                const int first = (wave - grp) & (NW - 1);
                grp += ngrp;
                if (first >= ngrp) continue;
                GEntry e = {};
                if (lane < n) e = ga.ent[base + lane];
                for (int j0 = first * GU; j0 < n; j0 += NW * GU) {   // GU rows in flight per wave
                    float4 v[GU];
                    float ly[GU], lx[GU], mm[GU];
#pragma unroll
                    for (int u = 0; u < GU; ++u) {
                        const int j = min(j0 + u, n - 1);
                        const int s = __builtin_amdgcn_readlane(e.s, j);
                        ly[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e.ly), j));
                        lx[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e.lx), j));
                        mm[u] = ga.raw ? __int_as_float(__builtin_amdgcn_readlane(e.pad, j)) : 1.f;
                        int row = s, c_lo = 0, c_hi = C;
                        if (ga.dg > 1) {
                            const int prow = s / KD, kd = s - prow * KD;
                            const int dgi = kd / K;
                            row = prow * K + (kd - dgi * K);
                            c_lo = dgi * cpdg, c_hi = c_lo + cpdg;
                        }
                        const bool on = (j0 + u < n) && c < c_hi && c >= c_lo;
                        v[u] = *reinterpret_cast<const float4 *>(ga.gcol + (size_t)row * C + (c < C ? c : 0));
                        if (!on) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int u = 0; u < GU; ++u) {
                        const float hy = 1.f - ly[u], hx = 1.f - lx[u];
#pragma unroll
                        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                            for (int dx = 0; dx < 2; ++dx) {
                                const float w = ((dy ? ly[u] : hy) * (dx ? lx[u] : hx)) * mm[u];
                                S[dy][dx].x += w * v[u].x, S[dy][dx].y += w * v[u].y;
                                S[dy][dx].z += w * v[u].z, S[dy][dx].w += w * v[u].w;
                            }
                    }
                    if (ga.Hb != nullptr) {   // corner sums of the GU entries x 4 corners
                        float d[16];
                        unsigned okm = 0;   // (entry, corner) pairs whose pixel belongs to this block
#pragma unroll
                        for (int u = 0; u < GU; ++u)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                d[u * 4 + q] = v[u].x * xa[q >> 1][q & 1].x + v[u].y * xa[q >> 1][q & 1].y +
                                               v[u].z * xa[q >> 1][q & 1].z + v[u].w * xa[q >> 1][q & 1].w;
                                if (j0 + u < n && inb[q >> 1][q & 1]) okm |= 1u << (u * 4 + q);
                            }
                        const float tot = wave64_sum16(d, lane);
                        const int su = __shfl(e.s, min(j0 + ((lane >> 2) & 3), n - 1));   // sample id of entry lane >> 2
                        if (lane < 16 && ((okm >> lane) & 1u)) {
                            float *hp = ga.Hb + (size_t)su * 4 + (lane & 3);
                            *hp = cb == 0 ? tot : *hp + tot;
                        }
                    }
                }
            }
            // fold the anchor's corner sums into the block (a wave owns its LDS rows: no barrier)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx)
                    if (inb[dy][dx]) {
                        float4 *ap = &acc[(ai - 1 + dy) * GT + (aj - 1 + dx)][lane];
                        float4 t = *ap;
                        t.x += S[dy][dx].x, t.y += S[dy][dx].y, t.z += S[dy][dx].z, t.w += S[dy][dx].w;
                        *ap = t;
                    }
        }
        if constexpr (NW == 4) __syncthreads();
        if (wave == 0 && c < C) {
#pragma unroll
            for (int i = 0; i < GT; ++i)
#pragma unroll
                for (int j = 0; j < GT; ++j)
                    if (y0 + i < G.H && x0 + j < G.W) {
                        float4 r = acc[i * GT + j][lane];
                        if constexpr (NW == 4) {   // (w0 + w2) + (w1 + w3)
                            const float4 r1 = accs[1][i * GT + j][lane], r2 = accs[2][i * GT + j][lane],
                                         r3 = accs[3][i * GT + j][lane];
                            r.x = (r.x + r2.x) + (r1.x + r3.x), r.y = (r.y + r2.y) + (r1.y + r3.y);
                            r.z = (r.z + r2.z) + (r1.z + r3.z), r.w = (r.w + r2.w) + (r1.w + r3.w);
                        }
                        *reinterpret_cast<float4 *>(G.gx + ((size_t)(b * G.H + y0 + i) * G.W + x0 + j) * C + c) = r;
                    }
        }
        if constexpr (NW == 4) __syncthreads();   // the accumulators are reset for the next channel block
    }
}

// NW = 4: one workgroup per anchor (long lists: its four waves share the entries and meet in LDS); NW = 1: one WAVE per
// anchor, four anchors per workgroup, no LDS and no barrier (short lists: the tower launch has ~5 entries per anchor, so
// 46 k independent waves of three dependent loads each hide the memory latency that one wave walking the 25 lists of a
// pixel block pays 25 times in a row).  Anchors [a_begin, a_begin + a_count) of the S buffer.
template <int NW>
__global__ __launch_bounds__(256) void dcn_anchor_sum_kernel(const AnchorArgs ga, int a_begin, int a_count)
{
    constexpr int GU = 4;
    __shared__ float4 red[NW == 4 ? 2 : 1][NW == 4 ? 4 : 1][NW == 4 ? 64 : 1];
    const int lane = threadIdx.x & 63, wave_id = threadIdx.x >> 6;
    const int wave = NW == 4 ? wave_id : 0;
    const int C = ga.C, K = ga.K, KD = ga.KD, cpdg = C / ga.dg;
    const int aidx = NW == 4 ? xcd_remap(blockIdx.x, gridDim.x) : xcd_remap(blockIdx.x, gridDim.x) * 4 + wave_id;
    if (aidx >= a_count) return;
    const int ai = a_begin + aidx;
    int gi = 0;
    while (gi + 1 < ga.ng && ai >= ga.g[gi + 1].a0) ++gi;
    const AnchorGrp &G = ga.g[gi];
    const int an = G.abase + (ai - G.a0);
    const int lb = ga.start[an], le = ga.start[an + 1];
    // anchor -> its four corner pixels (ay + dy, ax + dx); anchors live on the (H + 1) x (W + 1) grid shifted by one
    const int la = ai - G.a0, ab = la / ((G.H + 1) * (G.W + 1)), arem = la - ab * (G.H + 1) * (G.W + 1);
    const int ay = arem / (G.W + 1) - 1, ax = arem % (G.W + 1) - 1;
    const int cb0 = ga.ncb > 1 ? (int)blockIdx.y * 256 : 0, cb1 = ga.ncb > 1 ? min(cb0 + 256, C) : C;
    float *const Hbase = ga.Hb != nullptr ? ga.Hb + (size_t)(ga.ncb > 1 ? blockIdx.y : 0) * ga.hb_slot : nullptr;
    for (int cb = cb0; cb < cb1; cb += 256) {
        const int c = cb + lane * 4;
        float4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 xq[2][2];
        bool qin[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                qin[i][j] = ay + i >= 0 && ay + i < G.H && ax + j >= 0 && ax + j < G.W;
                xq[i][j] = (ga.Hb != nullptr && qin[i][j] && c < C)
                               ? *reinterpret_cast<const float4 *>(G.x + ((size_t)(ab * G.H + ay + i) * G.W + ax + j) * C + c)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        for (int base = lb; base < le; base += 64) {
            const int n = min(64, le - base);
            GEntry e = {};
            if (lane < n) e = ga.ent[base + lane];
            for (int j0 = wave * GU; j0 < n; j0 += NW * GU) {   // wave w: groups w, w + NW, ... of GU entries
                float4 v[GU];
                float ly[GU], lx[GU], mm[GU];
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    const int j = min(j0 + u, n - 1);
                    const int s = __builtin_amdgcn_readlane(e.s, j);
                    ly[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e.ly), j));
                    lx[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e.lx), j));
                    mm[u] = ga.raw ? __int_as_float(__builtin_amdgcn_readlane(e.pad, j)) : 1.f;
                    int row = s, c_lo = 0, c_hi = C;
                    if (ga.dg > 1) {
                        const int prow = s / KD, kd = s - prow * KD;
                        const int dgi = kd / K;
                        row = prow * K + (kd - dgi * K);
                        c_lo = dgi * cpdg, c_hi = c_lo + cpdg;
                    }
                    const bool on = (j0 + u < n) && c < c_hi && c >= c_lo;
                    v[u] = *reinterpret_cast<const float4 *>(ga.gcol + (size_t)row * C + (c < C ? c : 0));
                    if (!on) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    const float hy = 1.f - ly[u], hx = 1.f - lx[u];
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx) {
                            const float w = ((dy ? ly[u] : hy) * (dx ? lx[u] : hx)) * mm[u];
                            acc[dy][dx].x += w * v[u].x, acc[dy][dx].y += w * v[u].y;
                            acc[dy][dx].z += w * v[u].z, acc[dy][dx].w += w * v[u].w;
                        }
                }
                if (ga.Hb != nullptr) {
                    float d[16];
                    unsigned okm = 0;
#pragma unroll
                    for (int u = 0; u < GU; ++u)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            d[u * 4 + q] = v[u].x * xq[q >> 1][q & 1].x + v[u].y * xq[q >> 1][q & 1].y +
                                           v[u].z * xq[q >> 1][q & 1].z + v[u].w * xq[q >> 1][q & 1].w;
                            if (j0 + u < n && qin[q >> 1][q & 1]) okm |= 1u << (u * 4 + q);
                        }
                    const float tot = wave64_sum16(d, lane);
                    const int su = __shfl(e.s, min(j0 + ((lane >> 2) & 3), n - 1));
                    if (lane < 16 && ((okm >> lane) & 1u)) {
                        float *hp = Hbase + (size_t)su * 4 + (lane & 3);
                        *hp = cb == cb0 ? tot : *hp + tot;
                    }
                }
            }
        }
        // (w0 + w2) + (w1 + w3): fixed order
        auto put = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) red[slot][q][lane] = acc[q >> 1][q & 1];
        };
        auto add = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 r = red[slot][q][lane];
                acc[q >> 1][q & 1].x += r.x, acc[q >> 1][q & 1].y += r.y, acc[q >> 1][q & 1].z += r.z, acc[q >> 1][q & 1].w += r.w;
            }
        };
        if constexpr (NW == 4) {
            if (wave >= 2) put(wave - 2);
            __syncthreads();
            if (wave < 2) add(wave);
            __syncthreads();
            if (wave == 1) put(0);
            __syncthreads();
            if (wave == 0) add(0);
        }
        if (wave == 0 && c < C) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4 *>(ga.S + ((size_t)ai * 4 + q) * C + c) = acc[q >> 1][q & 1];
        }
        if constexpr (NW == 4) __syncthreads();
    }
}

// one wave per pixel of a long-list group
__global__ __launch_bounds__(256) void dcn_anchor_combine_kernel(const AnchorArgs ga, int npix)
{
    const int lane = threadIdx.x & 63;
    const int pi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pi >= npix) return;
    int gi = 0, p0 = 0;
    while (gi + 1 < ga.ng && pi >= p0 + ga.g[gi].B * ga.g[gi].H * ga.g[gi].W) p0 += ga.g[gi].B * ga.g[gi].H * ga.g[gi].W, ++gi;
    const AnchorGrp &G = ga.g[gi];
    const int lp = pi - p0;
    const int b = lp / (G.H * G.W), rem = lp - b * G.H * G.W;
    const int y = rem / G.W, x = rem - y * G.W;
    // anchor (ay, ax) (= floor of a sample position, >= -1) lives at row ay + 1, column ax + 1 of the (H + 1) x (W + 1) grid;
    // pixel (y, x) is corner (dy, dx) of anchor (y - dy, x - dx)
    auto S = [&](int ay, int ax, int q) -> const float * {
        return ga.S + ((size_t)(G.a0 + (b * (G.H + 1) + ay + 1) * (G.W + 1) + ax + 1) * 4 + q) * ga.C;
    };
    for (int c = lane * 4; c < ga.C; c += 256) {
        const float4 s11 = *reinterpret_cast<const float4 *>(S(y - 1, x - 1, 3) + c);
        const float4 s10 = *reinterpret_cast<const float4 *>(S(y - 1, x, 2) + c);
        const float4 s01 = *reinterpret_cast<const float4 *>(S(y, x - 1, 1) + c);
        const float4 s00 = *reinterpret_cast<const float4 *>(S(y, x, 0) + c);
        float4 r;
        r.x = ((s11.x + s10.x) + s01.x) + s00.x, r.y = ((s11.y + s10.y) + s01.y) + s00.y;
        r.z = ((s11.z + s10.z) + s01.z) + s00.z, r.w = ((s11.w + s10.w) + s01.w) + s00.w;
        *reinterpret_cast<float4 *>(G.gx + ((size_t)(b * G.H + y) * G.W + x) * ga.C + c) = r;
    }
}

// (Round 4 tried the gather per PIXEL instead -- one wave per input pixel walks the lists of the four anchors it is a
// corner of and writes grad_input straight from registers: no S round trip (0.37 GB per tower launch), each row read four
// times (L2).  It lost: tower backward 638 vs 600 us, pyramid 2031 vs 1603 us (profiles/r4_pixel_gather.txt); removed.)
// grad_offset / grad_mask from the corner sums Hb[sample][4] the gather pass left (kernel.cu:973-1044): one thread per
// sample.  A corner that lies outside the map has no list entry and no sum: its flag bit selects zero.
__global__ void dcn_offgrad_kernel(const DcnArgs a, int nsamples, const float4 *__restrict__ Hb, int ncb)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsamples) return;
    const int K = a.kh * a.kw, KD = K * a.dg;
    const int prow = s / KD, kd = s - prow * KD;
    const int dgi = kd / K, k = kd - dgi * K;
    const Lvl &L = find_level_by_row(a, prow);
    if (L.goff == nullptr && L.gmsk == nullptr) return;
    const Tap tp = a.gtap[(size_t)kd * a.gtap_rows + prow];
    float4 h = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tp.flags) {
        h = Hb[s];
        for (int y = 1; y < ncb; ++y) {   // (the channel blocks of AnchorArgs::ncb, in block order)
            const float4 t = Hb[(size_t)y * nsamples + s];
            h.x += t.x, h.y += t.y, h.z += t.z, h.w += t.w;
        }
    }
    const float hy = 1.f - tp.ly, hx = 1.f - tp.lx;
    const float v00 = (tp.flags & 1) ? h.x : 0.f, v01 = (tp.flags & 2) ? h.y : 0.f;
    const float v10 = (tp.flags & 4) ? h.z : 0.f, v11 = (tp.flags & 8) ? h.w : 0.f;
    const float dy = hx * (v10 - v00) + tp.lx * (v11 - v01);
    const float dx = hy * (v01 - v00) + tp.ly * (v11 - v10);
    const float bil = hy * hx * v00 + hy * tp.lx * v01 + tp.ly * hx * v10 + tp.ly * tp.lx * v11;
    const int pix = prow - L.prow0;
    const int HWo = L.Ho * L.Wo;
    const int b = pix / HWo, rem = pix - b * HWo;
    const int ho = rem / L.Wo, wo = rem - ho * L.Wo;
    if (L.goff) {
        float *op = L.goff + (size_t)b * L.osb + (size_t)ho * L.osh + (size_t)wo * L.osw;
        op[(size_t)(dgi * 2 * K + 2 * k) * L.osc] = tp.m * dy;
        op[(size_t)(dgi * 2 * K + 2 * k + 1) * L.osc] = tp.m * dx;
    }
    if (L.gmsk) {
        float gmv = bil;
        if (a.msig) gmv *= tp.m * (1.f - tp.m);
        L.gmsk[(size_t)b * L.msb + (size_t)(dgi * K + k) * L.msc + (size_t)ho * L.msh + (size_t)wo * L.msw] = gmv;
    }
}

}  // namespace lsn
