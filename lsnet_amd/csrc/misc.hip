// Sigmoid focal loss, NMS and the MFMA self-test of liblsnet_hip.so.
#include "common.h"

namespace lsn {

// ---------------------------------------------------------------------------------------------
// Sigmoid focal loss -- same formulas as sigmoid_focal_loss_cuda.cu:24-97:
//   FL  = -[t==d] a (1-p)^g log(max(p,FLT_MIN)) - [t!=d, t>=0] (1-a) p^g log(1-p)
//   log(1-p) in the stable form  -x[x>=0] - log(1 + exp(x - 2x[x>=0])).
// HBM-bound elementwise kernels: one float4 of logits per lane when C % 4 == 0.
// ---------------------------------------------------------------------------------------------
#define LSN_FLT_MIN 1.17549435e-38f

__device__ __forceinline__ float fl_forward_one(float x, bool pos, bool neg, float gamma, float alpha)
{
    const float p = 1.f / (1.f + expf(-x));
    const float xp = x >= 0.f ? 1.f : 0.f;
    const float log1mp = -x * xp - logf(1.f + expf(x - 2.f * x * xp));
    float l = 0.f;
    if (pos) l += -alpha * powf(1.f - p, gamma) * logf(fmaxf(p, LSN_FLT_MIN));
    if (neg) l += -(1.f - alpha) * powf(p, gamma) * log1mp;
    return l;
}

__device__ __forceinline__ float fl_backward_one(float x, bool pos, bool neg, float gamma, float alpha)
{
    const float p = 1.f / (1.f + expf(-x));
    const float xp = x >= 0.f ? 1.f : 0.f;
    const float log1mp = -x * xp - logf(1.f + expf(x - 2.f * x * xp));
    float v = 0.f;
    if (pos) v += -alpha * powf(1.f - p, gamma) * (1.f - p - p * gamma * logf(fmaxf(p, LSN_FLT_MIN)));
    if (neg) v += -(1.f - alpha) * powf(p, gamma) * (log1mp * (1.f - p) * gamma - p);
    return v;
}

__global__ void focal_fwd_kernel(const float *__restrict__ logits, const int64_t *__restrict__ targets,
                                 float *__restrict__ losses, int N, int C, float gamma, float alpha)
{
    const size_t total = (size_t)N * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / C), d = (int)(i - (size_t)n * C);
        const int t = (int)targets[n];
        losses[i] = fl_forward_one(logits[i], t == d, (t >= 0) & (t != d), gamma, alpha);
    }
}

__global__ void focal_bwd_kernel(const float *__restrict__ logits, const int64_t *__restrict__ targets,
                                 const float *__restrict__ d_losses, float *__restrict__ d_logits, int N, int C,
                                 float gamma, float alpha)
{
    const size_t total = (size_t)N * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / C), d = (int)(i - (size_t)n * C);
        const int t = (int)targets[n];
        d_logits[i] = fl_backward_one(logits[i], t == d, (t >= 0) & (t != d), gamma, alpha) * d_losses[i];
    }
}

__device__ __forceinline__ float block_sum_256(float v)
{
    __shared__ float part[4];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    return part[0] + part[1] + part[2] + part[3];
}

// loss_sum = sum_n w[n] * sum_c FL(n,c).  At most 1024 blocks (round 5; 256 until then: 41 transcendental-heavy elements per
// thread at the P3 level, 19 us per launch for 11 MB -- profiles/r5_pmc_hbm_stream.txt); each leaves its sum in part[] and takes a ticket, the block
// that draws the last one adds the partials up with the same fixed tree: no floating-point atomics, the same bits on
// every run (round 2: one fp32 atomic per block).
__global__ void focal_sum_kernel(const float *__restrict__ logits, const int64_t *__restrict__ targets,
                                 const float *__restrict__ weight, float *loss_sum, int N, int C, float gamma,
                                 float alpha, float *part, unsigned *ticket)
{
    const size_t total = (size_t)N * C;
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / C), d = (int)(i - (size_t)n * C);
        const int t = (int)targets[n];
        const float w = weight ? weight[n] : 1.f;
        s += w * fl_forward_one(logits[i], t == d, (t >= 0) & (t != d), gamma, alpha);
    }
    s = block_sum_256(s);
    __shared__ int last;
    if (threadIdx.x == 0) {
        store_agent(part + blockIdx.x, s);   // (common.h: agent-scope hand-over instead of a fence pair)
        wait_stores();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    float v = 0.f;   // up to four partials per thread, in block order
    for (unsigned b = threadIdx.x; b < gridDim.x; b += 256) v += load_agent(part + b);
    __syncthreads();   // (block_sum_256 reuses its LDS words)
    const float tot = block_sum_256(v);
    if (threadIdx.x == 0) {
        *loss_sum = tot;
        *ticket = 0;   // ready for the next launch on this stream
    }
}

// d_logits = (*scale) * w[n] * dFL/dx
__global__ void focal_bwd_w_kernel(const float *__restrict__ logits, const int64_t *__restrict__ targets,
                                   const float *__restrict__ weight, const float *__restrict__ scale,
                                   float *__restrict__ d_logits, int N, int C, float gamma, float alpha)
{
    const size_t total = (size_t)N * C;
    const float sc = *scale;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / C), d = (int)(i - (size_t)n * C);
        const int t = (int)targets[n];
        const float w = weight ? weight[n] : 1.f;
        d_logits[i] = sc * w * fl_backward_one(logits[i], t == d, (t >= 0) & (t != d), gamma, alpha);
    }
}

// ---- per-LEVEL sums over LSHead's concatenated pixel rows ----
// The head keeps the levels of an image back to back: row r = b * nall + i, level l owns i in [start[l], start[l + 1]).  The
// reference sums every loss per level (lsnet_head.py:1021-1270 loss_single through multi_apply): per level a slice copy, a sum
// kernel, a division and a multiplication, forward and backward -- ~70 launches of a few microseconds per step, each with its
// dispatch latency on the chain between forward and backward.  Here all levels are one launch and the sums come back as (L,).
constexpr int LV_MAX = 8;       // levels
constexpr int LV_RANGES = 64;   // (image, level) ranges of one launch
struct LevelTab {
    int B, nall, L;
    int start[LV_MAX + 1];      // first row of level l inside an image
    int blk0[LV_RANGES + 1];    // focal: first block of range k = b * L + l
};
__device__ __forceinline__ int level_of(const LevelTab &t, int i)
{
    int l = 0;
    while (l + 1 < t.L && i >= t.start[l + 1]) ++l;
    return l;
}

// sums[l] = sum over the rows of level l (all images) of w[n] * sum_c FL(n, c).  A block works inside ONE (image, level) range,
// leaves its sum in part[] and takes a ticket; the last block adds, per level, the partials of that level's blocks in block order.
__global__ void focal_level_sums_kernel(const float *__restrict__ logits, const int64_t *__restrict__ targets,
                                        const float *__restrict__ weight, float *sums, int C, float gamma, float alpha,
                                        float *part, unsigned *ticket, const LevelTab t)
{
    int k = 0;
    while ((int)blockIdx.x >= t.blk0[k + 1]) ++k;
    const int b = k / t.L, l = k - b * t.L;
    const int r0 = b * t.nall + t.start[l];
    const int nb = t.blk0[k + 1] - t.blk0[k], bi = blockIdx.x - t.blk0[k];
    const size_t total = (size_t)(t.start[l + 1] - t.start[l]) * C, base = (size_t)r0 * C;
    float s = 0.f;
    for (size_t i = bi * (size_t)256 + threadIdx.x; i < total; i += (size_t)nb * 256) {
        const int n = (int)(i / C), d = (int)(i - (size_t)n * C);
        const int tg = (int)targets[r0 + n];
        const float w = weight ? weight[r0 + n] : 1.f;
        s += w * fl_forward_one(logits[base + i], tg == d, (tg >= 0) & (tg != d), gamma, alpha);
    }
    s = block_sum_256(s);
    __shared__ int last;
    if (threadIdx.x == 0) {
        store_agent(part + blockIdx.x, s);
        wait_stores();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    for (int lv = 0; lv < t.L; ++lv) {
        float v = 0.f;
        for (int img = 0; img < t.B; ++img) {
            const int kk = img * t.L + lv;
            for (int blk = t.blk0[kk] + threadIdx.x; blk < t.blk0[kk + 1]; blk += 256) v += load_agent(part + blk);
        }
        __syncthreads();   // (block_sum_256 reuses its LDS words)
        const float tot = block_sum_256(v);
        if (threadIdx.x == 0) sums[lv] = tot;
    }
    if (threadIdx.x == 0) *ticket = 0;
}

// d_logits = scale[level(n)] * w[n] * dFL/dx
__global__ void focal_bwd_w_levels_kernel(const float *__restrict__ logits, const int64_t *__restrict__ targets,
                                          const float *__restrict__ weight, const float *__restrict__ scale,
                                          float *__restrict__ d_logits, int N, int C, float gamma, float alpha, const LevelTab t)
{
    const size_t total = (size_t)N * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / C), d = (int)(i - (size_t)n * C);
        const int tg = (int)targets[n];
        const float sc = scale[level_of(t, n % t.nall)] * (weight ? weight[n] : 1.f);
        d_logits[i] = sc * fl_backward_one(logits[i], tg == d, (tg >= 0) & (tg != d), gamma, alpha);
    }
}

// sums[l] = sum of rows[b * nall + i] over the images and the rows i of level l: one block per level, a fixed order
__global__ __launch_bounds__(1024) void level_sums_kernel(const float *__restrict__ rows, float *sums, const LevelTab t)
{
    const int l = blockIdx.x;
    float v = 0.f;
    for (int b = 0; b < t.B; ++b)
        for (int i = t.start[l] + threadIdx.x; i < t.start[l + 1]; i += 1024) v += rows[(size_t)b * t.nall + i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    __shared__ float part[16];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int w = 0; w < 16; ++w) tot += part[w];
        sums[l] = tot;
    }
}

// out[r] = g[level(r)]: the gradient of level_sums
__global__ void level_expand_kernel(const float *__restrict__ g, float *__restrict__ out, int N, const LevelTab t)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < N) out[r] = g[level_of(t, r % t.nall)];
}

static int ew_grid(size_t total) { size_t g = (total + 255) / 256; return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g)); }

// ---------------------------------------------------------------------------------------------
// NMS.  Same two-phase idea as nms_kernel.cu (64x64 IoU tiles -> 64-bit suppression masks) but the
// sequential sweep also runs on the device (one wavefront), so nothing is copied to the host.
// IoU arithmetic uses explicitly rounded single operations (no FMA contraction) so that the keep
// set is bit-identical to the CPU reference (nms_cpu.cpp:21-63).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool iou_gt(const float *a, const float *b, float thr)
{
    const float xx1 = fmaxf(a[0], b[0]), yy1 = fmaxf(a[1], b[1]);
    const float xx2 = fminf(a[2], b[2]), yy2 = fminf(a[3], b[3]);
    const float w = fmaxf(0.f, __fsub_rn(xx2, xx1)), h = fmaxf(0.f, __fsub_rn(yy2, yy1));
    const float inter = __fmul_rn(w, h);
    const float aa = __fmul_rn(__fsub_rn(a[2], a[0]), __fsub_rn(a[3], a[1]));
    const float ab = __fmul_rn(__fsub_rn(b[2], b[0]), __fsub_rn(b[3], b[1]));
    const float ovr = __fdiv_rn(inter, __fsub_rn(__fadd_rn(aa, ab), inter));
    return ovr > thr;
}

__global__ void nms_mask_kernel(const float *__restrict__ dets, const int64_t *__restrict__ order, int n,
                                float thr, unsigned long long *__restrict__ mask)
{
    const int rb = blockIdx.y, cb = blockIdx.x;
    const int nb = gridDim.x;
    __shared__ float cbox[64][4];
    const int cj = cb * 64 + threadIdx.x;
    if (cj < n) {
        const float *d = dets + (size_t)order[cj] * 5;
        cbox[threadIdx.x][0] = d[0];
        cbox[threadIdx.x][1] = d[1];
        cbox[threadIdx.x][2] = d[2];
        cbox[threadIdx.x][3] = d[3];
    }
    __syncthreads();
    const int ri = rb * 64 + threadIdx.x;
    if (ri >= n) return;
    unsigned long long bits = 0ull;
    if (cb >= rb) {
        const float *d = dets + (size_t)order[ri] * 5;
        const float rbox[4] = {d[0], d[1], d[2], d[3]};
        const int ncol = min(64, n - cb * 64);
        const int start = (cb == rb) ? threadIdx.x + 1 : 0;
        for (int j = start; j < ncol; ++j)
            if (iou_gt(rbox, cbox[j], thr)) bits |= 1ull << j;
    }
    mask[(size_t)ri * nb + cb] = bits;
}

__global__ void nms_sweep_kernel(const unsigned long long *__restrict__ mask, const int64_t *__restrict__ order,
                                 int n, int nb, int64_t *__restrict__ keep, int64_t *__restrict__ num_keep)
{
    extern __shared__ unsigned long long removed[];  // nb words
    for (int w = threadIdx.x; w < nb; w += blockDim.x) removed[w] = 0ull;
    __syncthreads();
    int nk = 0;
    for (int i = 0; i < n; ++i) {
        const bool dead = (removed[i >> 6] >> (i & 63)) & 1ull;  // uniform across the block
        if (!dead) {
            if (threadIdx.x == 0) keep[nk] = order[i];
            ++nk;
            for (int w = (i >> 6) + threadIdx.x; w < nb; w += blockDim.x) removed[w] |= mask[(size_t)i * nb + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *num_keep = nk;
}

// ---------------------------------------------------------------------------------------------
// MFMA self-test: D = A(MxK) * B(KxN), one wavefront per 32x32 (variant 0) or 16x16 (variant 1)
// tile, operands read straight from global memory with the fragment maps of common.h.
// ---------------------------------------------------------------------------------------------
__global__ void selftest32_kernel(const float *A, const float *B, float *D, int M, int N, int K)
{
    const int lane = threadIdx.x, i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 2) {
        const int k = k0 + (lane >> 5), i = i0 + (lane & 31), j = j0 + (lane & 31);
        const float a = (i < M && k < K) ? A[(size_t)i * K + k] : 0.f;
        const float b = (j < N && k < K) ? B[(size_t)k * N + j] : 0.f;
        acc = mfma32(a, b, acc);
    }
    for (int r = 0; r < 16; ++r) {
        const int i = i0 + mfma32_row(r, lane), j = j0 + (lane & 31);
        if (i < M && j < N) D[(size_t)i * N + j] = acc[r];
    }
}

__global__ void selftest16_kernel(const float *A, const float *B, float *D, int M, int N, int K)
{
    const int lane = threadIdx.x, i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int k = k0 + (lane >> 4), i = i0 + (lane & 15), j = j0 + (lane & 15);
        const float a = (i < M && k < K) ? A[(size_t)i * K + k] : 0.f;
        const float b = (j < N && k < K) ? B[(size_t)k * N + j] : 0.f;
        acc = mfma16(a, b, acc);
    }
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + 4 * (lane >> 4) + r, j = j0 + (lane & 15);
        if (i < M && j < N) D[(size_t)i * N + j] = acc[r];
    }
}


// ---------------------------------------------------------------------------------------------
// The k smallest (or largest) entries of every COLUMN of a row-major (P, G) matrix, per row segment: what
// `torch.topk(dist[start:start + n], k, dim=0, largest=False)` returns for the assigners' (points x gts) distance
// matrices (centroid_assigner.py:74, atss_assigner.py:103-111) -- there as one single-block radix select per column and
// call (37 - 100 us each, 12 calls per step); here one workgroup per (column, segment), all segments in one launch.
// A column segment is read once into LDS as order-preserving 32-bit keys (NaN above everything, as torch orders it);
// k rounds of a workgroup-wide minimum over (key, row) pairs then emit the entries in ascending order, equal values in
// ascending row order (torch leaves the order of equal values unspecified).
// ---------------------------------------------------------------------------------------------
struct TopkSegs {
    int start[8], len[8];
};

__device__ __forceinline__ unsigned topk_key(float v, bool largest)
{
    unsigned u = __float_as_uint(v);
    u = (v != v) ? 0xffffffffu                                       // NaN: the largest value, as torch orders it
                 : ((u & 0x80000000u) ? ~u : (u | 0x80000000u));     // ascending float order as unsigned order
    return largest ? ~u : u;
}

__global__ __launch_bounds__(1024) void topk_cols_kernel(const float *__restrict__ x, int ldx, int G, TopkSegs segs, int k, int largest,
                                                         float *__restrict__ vals, long long *__restrict__ idx, int cap)
{
    // 1024 threads: the column is read with a stride of G floats -- one element per 4 x G bytes, nothing to coalesce -- so
    // the read is a latency problem: 16 waves with 8 independent loads each in flight (256 threads and a rolled loop took
    // 75 us for the centroid assigner's 22 400-row columns, profiles/r4c_kernel_stats.txt)
    extern __shared__ unsigned keys[];      // min(len, cap) keys of the segment's column
    __shared__ unsigned long long red[16];
    const int g = blockIdx.x, sg = blockIdx.y;
    const int start = segs.start[sg], n = segs.len[sg];
    const int tid = threadIdx.x;
    const float *col = x + (size_t)start * ldx + g;
    const int nc = n < cap ? n : cap;
    for (int i0 = tid; i0 < nc; i0 += 8 * 1024) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = i0 + u * 1024 < nc ? col[(size_t)(i0 + u * 1024) * ldx] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + u * 1024 < nc) keys[i0 + u * 1024] = topk_key(v[u], largest != 0);
    }
    __syncthreads();
    unsigned long long last = 0;            // (key << 32 | row) of the previous pick; picks are strictly increasing
    bool first = true;
    for (int r = 0; r < k; ++r) {
        unsigned long long best = ~0ull;
        for (int i = tid; i < n; i += 1024) {
            const unsigned key = i < nc ? keys[i] : topk_key(col[(size_t)i * ldx], largest != 0);
            const unsigned long long c = ((unsigned long long)key << 32) | (unsigned)i;
            if ((first || c > last) && c < best) best = c;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long t = __shfl_xor(best, o);
            best = t < best ? t : best;
        }
        if ((tid & 63) == 0) red[tid >> 6] = best;
        __syncthreads();
        best = red[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) best = red[w] < best ? red[w] : best;
        __syncthreads();
        if (tid == 0) {
            const size_t o = ((size_t)sg * k + r) * G + g;
            if (best == ~0ull) {            // fewer than k rows: cannot happen (checked by the host)
                vals[o] = 0.f, idx[o] = -1;
            } else {
                const int row = (int)(best & 0xffffffffu);
                vals[o] = col[(size_t)row * ldx];
                idx[o] = start + row;
            }
        }
        last = best, first = false;
    }
}

// ---------------------------------------------------------------------------------------------
// The offsets LSHead hands to its pyramid convolutions (lsnet_head.py:622-638): a level's offset field is rescaled IN
// PLACE while the three source levels are visited, so the three fields are off m1, (off m1) m2, ((off m1) m2) m3 with
// m_k = (scale_h, scale_w) of source k on the (y, x) channel pairs.  In torch that is 3 multiplications per level forward
// and 3 multiplications + 2 additions backward -- 15 + 25 launches of a few microseconds per step; here one launch per
// direction over all levels.  Every product and sum is a separately rounded fp32 operation (no fma): bit-identical to the
// operator sequence.
// ---------------------------------------------------------------------------------------------
struct ChainArgs {
    lsn_offset_chain_level lv[8];
    long long first[9];   // first element of every level in the launch-wide numbering
    int n, C;
};

template <bool BWD>
__global__ __launch_bounds__(256) void offset_chain_kernel(const ChainArgs a)
{
    const long long total = a.first[a.n];
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        int l = 0;
        while (l + 1 < a.n && e >= a.first[l + 1]) ++l;
        const lsn_offset_chain_level &L = a.lv[l];
        const long long i = e - a.first[l];                 // element of the dense (B, H, W, C) tensors
        const bool x = ((int)(i % a.C) & 1) != 0;           // channels: y0 x0 y1 x1 ...
        const float m0 = x ? L.mw[0] : L.mh[0], m1 = x ? L.mw[1] : L.mh[1], m2 = x ? L.mw[2] : L.mh[2];
        if (!BWD) {
            const long long b = i / L.per_image;
            const float v = L.off[b * L.off_image_pitch + (i - b * L.per_image)];
            const float o0 = v * m0, o1 = o0 * m1, o2 = o1 * m2;
            L.out[0][i] = o0, L.out[1][i] = o1, L.out[2][i] = o2;
        } else {
            // products through an asm statement: hipcc's __fmul_rn / __fadd_rn are plain operators, and -ffp-contract=fast fuses
            // them with the additions (and does not honour `#pragma clang fp contract(off)`)
            float g0 = L.gout[0] ? L.gout[0][i] : 0.f, g1 = L.gout[1] ? L.gout[1][i] : 0.f, g2 = L.gout[2] ? L.gout[2][i] : 0.f;
            if (L.gout[3]) g0 = L.gout[0] ? g0 + L.gout[3][i] : L.gout[3][i];   // (the accumulation autograd does for two consumers)
            if (L.gout[4]) g1 = L.gout[1] ? g1 + L.gout[4][i] : L.gout[4][i];
            if (L.gout[5]) g2 = L.gout[2] ? g2 + L.gout[5][i] : L.gout[5][i];
            float p2, p1;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p2) : "v"(g2), "v"(m2));
            const float t1 = g1 + p2;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p1) : "v"(t1), "v"(m1));
            const float t0 = g0 + p1;
            L.goff[i] = t0 * m0;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Gradient clipping + SGD step (mmcv/runner/hooks/optimizer.py:8-28: clip_grad_norm_(params, max_norm, 2) then
// torch.optim.SGD.step()) over every parameter tensor in three launches: partial sums of squares, their ordered sum -> the
// total norm and the clip coefficient (device scalars, no host read), the update.  Per element, with every operation
// rounded as the torch operator sequence rounds it (multiplication by the coefficient; grad.add(param, alpha = wd) and
// param.add_(buf, alpha = -lr) as fused multiply-adds, which is how ATen's `a + alpha * b` compiles; buf.mul_(momentum)
// and the addition behind it separately):
//     g' = g c,   d = fma(wd, p, g'),   buf = (buf m) + d,   p = fma(-lr, buf, p)
// A zero-filled momentum buffer gives the first step's buf = d, as torch's clone of the gradient does.
// ---------------------------------------------------------------------------------------------
struct SgdArgs {
    const lsn_sgd_tensor *t;      // device table, sorted by `start`
    int n;
    long long chunks;             // 1024-float4 chunks of all tensors
    lsn_sgd_group g[8];
    float max_norm;
    double *partial;              // [gridDim.x] sums of squares
    float *stats;                 // [0] total norm, [1] clip coefficient (1 without clipping)
};

__device__ __forceinline__ int sgd_find(const SgdArgs &a, long long chunk)
{
    int lo = 0, hi = a.n - 1;     // last tensor with first_chunk <= chunk (wave-uniform: a workgroup owns a chunk)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.t[mid].first_chunk <= chunk) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(256) void sgd_sqnorm_kernel(const SgdArgs a)
{
    __shared__ double red[4];
    double acc = 0.0;
    for (long long c = blockIdx.x; c < a.chunks; c += gridDim.x) {
        const lsn_sgd_tensor T = a.t[sgd_find(a, c)];
        const long long e0 = (c - T.first_chunk) * 4096 + threadIdx.x * 4;
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long e = e0 + u * 1024;
            if (e + 3 < T.numel) {
                const float4 v = *reinterpret_cast<const float4 *>(T.grad + e);
                s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            } else {
                for (long long q = e; q < T.numel && q < e + 4; ++q) s += T.grad[q] * T.grad[q];
            }
        }
        acc += (double)s;
    }
    for (int m = 1; m < 64; m <<= 1) acc += __shfl_xor(acc, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) a.partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sgd_coef_kernel(const SgdArgs a, int nparts)
{
    __shared__ double red[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += a.partial[i];
    for (int m = 1; m < 64; m <<= 1) acc += __shfl_xor(acc, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float total = (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
        const float coef = a.max_norm / (total + 1e-6f);      // clip_grad_norm_: clip_coef, clamped to <= 1
        a.stats[0] = total;
        // clamp(coef, max = 1) as torch computes it: a NaN norm gives a NaN coefficient and every gradient becomes NaN
        // (`coef < 1 ? coef : 1` turned it into 1 and the step went ahead on the unclipped gradients)
        a.stats[1] = (coef < 1.f || coef != coef) ? coef : 1.f;
    }
}

__global__ __launch_bounds__(256) void sgd_step_kernel(const SgdArgs a)
{
    const float coef = a.max_norm > 0.f ? a.stats[1] : 1.f;
    const bool scaled = coef != 1.f;      // torch scales the gradients in place: keep p.grad as the reference leaves it
    for (long long c = blockIdx.x; c < a.chunks; c += gridDim.x) {
        const lsn_sgd_tensor T = a.t[sgd_find(a, c)];
        const lsn_sgd_group G = a.g[T.group];
        const float nlr = -G.lr;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long e = (c - T.first_chunk) * 4096 + u * 1024 + threadIdx.x * 4;
            if (e >= T.numel) continue;
            const int cnt = T.numel - e >= 4 ? 4 : (int)(T.numel - e);
            float g[4], p[4], b[4];
            if (cnt == 4) {
                const float4 gv = *reinterpret_cast<const float4 *>(T.grad + e), pv = *reinterpret_cast<const float4 *>(T.param + e);
                const float4 bv = *reinterpret_cast<const float4 *>(T.momentum_buf + e);
                g[0] = gv.x, g[1] = gv.y, g[2] = gv.z, g[3] = gv.w, p[0] = pv.x, p[1] = pv.y, p[2] = pv.z, p[3] = pv.w;
                b[0] = bv.x, b[1] = bv.y, b[2] = bv.z, b[3] = bv.w;
            } else {
                for (int q = 0; q < 4; ++q) g[q] = q < cnt ? T.grad[e + q] : 0.f, p[q] = q < cnt ? T.param[e + q] : 0.f, b[q] = q < cnt ? T.momentum_buf[e + q] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float gs, bm;
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(gs) : "v"(g[q]), "v"(coef));          // g' = g c          (rounded)
                const float d = G.weight_decay != 0.f ? __builtin_fmaf(G.weight_decay, p[q], gs) : gs;
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(bm) : "v"(b[q]), "v"(G.momentum));     // buf m             (rounded)
                float bn;
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(bn) : "v"(bm), "v"(d));               // + d               (rounded)
                g[q] = gs, b[q] = bn, p[q] = __builtin_fmaf(nlr, bn, p[q]);
            }
            if (cnt == 4) {
                *reinterpret_cast<float4 *>(T.param + e) = make_float4(p[0], p[1], p[2], p[3]);
                *reinterpret_cast<float4 *>(T.momentum_buf + e) = make_float4(b[0], b[1], b[2], b[3]);
                if (scaled) *reinterpret_cast<float4 *>(T.grad + e) = make_float4(g[0], g[1], g[2], g[3]);
            } else {
                for (int q = 0; q < cnt; ++q) {
                    T.param[e + q] = p[q], T.momentum_buf[e + q] = b[q];
                    if (scaled) T.grad[e + q] = g[q];
                }
            }
        }
    }
}

}  // namespace lsn

using namespace lsn;

extern "C" {

int lsn_sigmoid_focal_loss_forward(const float *logits, const int64_t *targets, float *losses, int N, int C,
                                   float gamma, float alpha, lsn_stream_t stream)
{
    LSN_CHECK(N >= 0 && C > 0, "invalid focal loss shape (%d, %d)", N, C);
    if (N == 0) return 0;
    hipLaunchKernelGGL(focal_fwd_kernel, dim3(ew_grid((size_t)N * C)), dim3(256), 0, stream, logits, targets,
                       losses, N, C, gamma, alpha);
    LSN_HIP(hipGetLastError());
    return 0;
}

int lsn_sigmoid_focal_loss_backward(const float *logits, const int64_t *targets, const float *d_losses,
                                    float *d_logits, int N, int C, float gamma, float alpha, lsn_stream_t stream)
{
    LSN_CHECK(N >= 0 && C > 0, "invalid focal loss shape (%d, %d)", N, C);
    if (N == 0) return 0;
    hipLaunchKernelGGL(focal_bwd_kernel, dim3(ew_grid((size_t)N * C)), dim3(256), 0, stream, logits, targets,
                       d_losses, d_logits, N, C, gamma, alpha);
    LSN_HIP(hipGetLastError());
    return 0;
}

// library-owned scratch of the two-stage focal sums (one stream at a time, like the other scratch buffers of the library)
static int focal_scratch(float **part_out, unsigned **ticket_out)
{
    static float *part = nullptr;
    static unsigned *ticket = nullptr;
    if (!part) {
        LSN_HIP(hipMalloc(reinterpret_cast<void **>(&part), 1024 * sizeof(float) + sizeof(unsigned)));
        lsn::lib_stat(lsn::STAT_MALLOCS, 1), lsn::lib_stat(lsn::STAT_HELD_BYTES, 1024 * sizeof(float) + sizeof(unsigned));
        ticket = reinterpret_cast<unsigned *>(part + 1024);
        LSN_HIP(hipMemset(ticket, 0, sizeof(unsigned)));
    }
    *part_out = part, *ticket_out = ticket;
    return 0;
}

static int level_tab(lsn::LevelTab &t, int B, int N_all, int L, const int *level_starts)
{
    LSN_CHECK(B >= 1 && L >= 1 && L <= lsn::LV_MAX && B * L <= lsn::LV_RANGES && level_starts, "level sums: %d images x %d levels", B, L);
    LSN_CHECK(level_starts[0] == 0 && level_starts[L] == N_all, "level sums: the levels must cover rows 0 .. %d", N_all);
    t.B = B, t.nall = N_all, t.L = L;
    for (int l = 0; l <= L; ++l) t.start[l] = level_starts[l];
    for (int l = 0; l < L; ++l) LSN_CHECK(t.start[l + 1] > t.start[l], "level sums: level %d is empty", l);
    return 0;
}

int lsn_sigmoid_focal_loss_sum(const float *logits, const int64_t *targets, const float *weight, float *loss_sum,
                               int N, int C, float gamma, float alpha, lsn_stream_t stream)
{
    LSN_CHECK(N >= 0 && C > 0, "invalid focal loss shape (%d, %d)", N, C);
    if (N == 0) {
        LSN_HIP(hipMemsetAsync(loss_sum, 0, sizeof(float), stream));
        return 0;
    }
    float *part = nullptr;
    unsigned *ticket = nullptr;
    if (int rc = focal_scratch(&part, &ticket)) return rc;
    int grid = ew_grid((size_t)N * C);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(focal_sum_kernel, dim3(grid), dim3(256), 0, stream, logits, targets, weight, loss_sum, N, C, gamma,
                       alpha, part, ticket);
    LSN_HIP(hipGetLastError());
    return 0;
}

int lsn_sigmoid_focal_loss_backward_weighted(const float *logits, const int64_t *targets, const float *weight,
                                             const float *scale, float *d_logits, int N, int C, float gamma,
                                             float alpha, lsn_stream_t stream)
{
    LSN_CHECK(N >= 0 && C > 0, "invalid focal loss shape (%d, %d)", N, C);
    if (N == 0) return 0;
    hipLaunchKernelGGL(focal_bwd_w_kernel, dim3(ew_grid((size_t)N * C)), dim3(256), 0, stream, logits, targets,
                       weight, scale, d_logits, N, C, gamma, alpha);
    LSN_HIP(hipGetLastError());
    return 0;
}

int lsn_sigmoid_focal_loss_level_sums(const float *logits, const int64_t *targets, const float *weight, float *loss_sums,
                                      int B, int N_all, int C, int L, const int *level_starts, float gamma, float alpha,
                                      lsn_stream_t stream)
{
    using namespace lsn;
    LSN_CHECK(C > 0 && logits && targets && loss_sums, "invalid focal loss arguments");
    LevelTab t;
    if (int rc = level_tab(t, B, N_all, L, level_starts)) return rc;
    float *part = nullptr;
    unsigned *ticket = nullptr;
    if (int rc = focal_scratch(&part, &ticket)) return rc;
    // ~16 elements per thread; coarser until the launch fits the 1024 partial sums
    for (long long per = 4096;; per *= 2) {
        int nb = 0;
        for (int k = 0; k < B * L; ++k) {
            const long long el = (long long)(t.start[k % L + 1] - t.start[k % L]) * C;
            t.blk0[k] = nb;
            nb += (int)((el + per - 1) / per);
        }
        t.blk0[B * L] = nb;
        if (nb <= 1024) break;
    }
    hipLaunchKernelGGL(focal_level_sums_kernel, dim3(t.blk0[B * L]), dim3(256), 0, stream, logits, targets, weight, loss_sums, C,
                       gamma, alpha, part, ticket, t);
    LSN_HIP(hipGetLastError());
    return 0;
}

int lsn_sigmoid_focal_loss_backward_levels(const float *logits, const int64_t *targets, const float *weight, const float *scales,
                                           float *d_logits, int B, int N_all, int C, int L, const int *level_starts,
                                           float gamma, float alpha, lsn_stream_t stream)
{
    using namespace lsn;
    LSN_CHECK(C > 0 && logits && targets && scales && d_logits, "invalid focal loss arguments");
    LevelTab t;
    if (int rc = level_tab(t, B, N_all, L, level_starts)) return rc;
    const int N = B * N_all;
    hipLaunchKernelGGL(focal_bwd_w_levels_kernel, dim3(ew_grid((size_t)N * C)), dim3(256), 0, stream, logits, targets, weight,
                       scales, d_logits, N, C, gamma, alpha, t);
    LSN_HIP(hipGetLastError());
    return 0;
}

int lsn_level_sums(const float *rows, float *sums, int B, int N_all, int L, const int *level_starts, lsn_stream_t stream)
{
    using namespace lsn;
    LSN_CHECK(rows && sums, "level sums: NULL argument");
    LevelTab t;
    if (int rc = level_tab(t, B, N_all, L, level_starts)) return rc;
    hipLaunchKernelGGL(level_sums_kernel, dim3(L), dim3(1024), 0, stream, rows, sums, t);
    LSN_HIP(hipGetLastError());
    return 0;
}

int lsn_level_expand(const float *g, float *out_rows, int B, int N_all, int L, const int *level_starts, lsn_stream_t stream)
{
    using namespace lsn;
    LSN_CHECK(g && out_rows, "level sums: NULL argument");
    LevelTab t;
    if (int rc = level_tab(t, B, N_all, L, level_starts)) return rc;
    const int N = B * N_all;
    hipLaunchKernelGGL(level_expand_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, g, out_rows, N, t);
    LSN_HIP(hipGetLastError());
    return 0;
}

int64_t lsn_nms_workspace_bytes(int n)
{
    if (n <= 0) return 8;
    const int64_t nb = (n + 63) / 64;
    return (int64_t)n * nb * 8;
}

int lsn_nms(const float *dets, const int64_t *order, int n, float iou_thr, int64_t *keep, int64_t *num_keep,
            void *workspace, lsn_stream_t stream)
{
    LSN_CHECK(n >= 0, "invalid number of boxes %d", n);
    if (n == 0) {
        LSN_HIP(hipMemsetAsync(num_keep, 0, sizeof(int64_t), stream));
        return 0;
    }
    const int nb = (n + 63) / 64;
    if ((size_t)nb * 8 > 64 * 1024) return fail(LSN_ERR_UNSUPPORTED, "nms: %d boxes exceed the sweep kernel's LDS", n);
    unsigned long long *mask = reinterpret_cast<unsigned long long *>(workspace);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(nb, nb), dim3(64), 0, stream, dets, order, n, iou_thr, mask);
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(64), (size_t)nb * 8, stream, mask, order, n, nb, keep,
                       num_keep);
    LSN_HIP(hipGetLastError());
    return 0;
}

int lsn_selftest_mfma(const float *A, const float *B, float *D, int M, int N, int K, int variant,
                      lsn_stream_t stream)
{
    LSN_CHECK(M > 0 && N > 0 && K > 0, "invalid GEMM shape");
    if (variant == 0)
        hipLaunchKernelGGL(selftest32_kernel, dim3(cdiv(N, 32), cdiv(M, 32)), dim3(64), 0, stream, A, B, D, M, N, K);
    else
        hipLaunchKernelGGL(selftest16_kernel, dim3(cdiv(N, 16), cdiv(M, 16)), dim3(64), 0, stream, A, B, D, M, N, K);
    LSN_HIP(hipGetLastError());
    return 0;
}

int lsn_topk_columns(const float *x, int P, int G, int ldx, int nseg, const int *seg_start, const int *seg_len, int k, int largest,
                     float *values, int64_t *indices, lsn_stream_t stream)
{
    LSN_CHECK(P >= 0 && G >= 0 && ldx >= G && k > 0, "topk: invalid shape (P %d, G %d, ldx %d, k %d)", P, G, ldx, k);
    LSN_CHECK(nseg >= 1 && nseg <= 8, "topk: %d segments (1 .. 8)", nseg);
    lsn::TopkSegs segs;
    for (int i = 0; i < nseg; ++i) {
        LSN_CHECK(seg_start[i] >= 0 && seg_len[i] >= k && seg_start[i] + seg_len[i] <= P,
                  "topk: segment %d = [%d, +%d) of %d rows with k = %d", i, seg_start[i], seg_len[i], P, k);
        segs.start[i] = seg_start[i], segs.len[i] = seg_len[i];
    }
    if (G == 0) return 0;
    int nmax = 0;
    for (int i = 0; i < nseg; ++i) nmax = seg_len[i] > nmax ? seg_len[i] : nmax;
    // keys in LDS: as many as the CURRENT device allows a workgroup to opt into (160 KB on gfx950 -> 36 K keys = 144 KB);
    // longer columns re-read the rest from L2.  The attribute is per device and per process: set it on every call (a
    // host-side table write) instead of once behind a process-wide flag.
    int dev = 0, lds_max = 0;
    LSN_HIP(hipGetDevice(&dev));
    LSN_HIP(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeSharedMemPerBlockOptin, dev));
    if (lds_max <= 0) LSN_HIP(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
    int cap_max = (lds_max - 4096) / 4;   // (the kernel's static LDS: reduction scratch)
    if (cap_max > 36 * 1024) cap_max = 36 * 1024;
    if (cap_max < 1024) return lsn::fail(LSN_ERR_UNSUPPORTED, "topk: device offers %d bytes of LDS per workgroup", lds_max);
    const int cap = nmax < cap_max ? nmax : cap_max;
    LSN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(lsn::topk_cols_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                cap_max * 4));
    hipLaunchKernelGGL(lsn::topk_cols_kernel, dim3(G, nseg), dim3(1024), (size_t)cap * 4, stream, x, ldx, G, segs, k, largest,
                       values, reinterpret_cast<long long *>(indices), cap);
    LSN_HIP(hipGetLastError());
    return 0;
}

static int offset_chain_launch(int n_levels, const lsn_offset_chain_level *levels, int C, bool bwd, lsn_stream_t stream)
{
    LSN_CHECK(n_levels >= 1 && n_levels <= 8 && levels && C >= 2 && C % 2 == 0, "offset chain: bad arguments");
    lsn::ChainArgs a;
    a.n = n_levels, a.C = C;
    long long tot = 0;
    for (int i = 0; i < n_levels; ++i) {
        const lsn_offset_chain_level &L = levels[i];
        LSN_CHECK(L.images > 0 && L.per_image > 0 && L.per_image % C == 0, "offset chain: level %d: bad sizes", i);
        if (bwd)
            LSN_CHECK(L.goff != nullptr, "offset chain: level %d: grad_offset is NULL", i);
        else
            LSN_CHECK(L.off && L.out[0] && L.out[1] && L.out[2] && L.off_image_pitch >= L.per_image, "offset chain: level %d: bad pointers", i);
        a.lv[i] = L;
        a.first[i] = tot;
        tot += (long long)L.images * L.per_image;
    }
    a.first[n_levels] = tot;
    const int blocks = (int)((tot + 255) / 256 < 2048 ? (tot + 255) / 256 : 2048);
    if (bwd)
        hipLaunchKernelGGL(lsn::offset_chain_kernel<true>, dim3(blocks), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(lsn::offset_chain_kernel<false>, dim3(blocks), dim3(256), 0, stream, a);
    LSN_HIP(hipGetLastError());
    return 0;
}

int lsn_offset_chain_forward(int n_levels, const lsn_offset_chain_level *levels, int C, lsn_stream_t stream)
{
    return offset_chain_launch(n_levels, levels, C, false, stream);
}

int lsn_offset_chain_backward(int n_levels, const lsn_offset_chain_level *levels, int C, lsn_stream_t stream)
{
    return offset_chain_launch(n_levels, levels, C, true, stream);
}

int64_t lsn_clip_sgd_workspace_bytes(void) { return 2048 * 8 + 64; }

int lsn_clip_sgd_step(int n_tensors, const lsn_sgd_tensor *tensors_dev, int64_t total_chunks, int n_groups,
                      const lsn_sgd_group *groups, float max_norm, void *workspace, float *stats, lsn_stream_t stream)
{
    LSN_CHECK(n_tensors >= 1 && tensors_dev && total_chunks >= 1 && n_groups >= 1 && n_groups <= 8 && groups && workspace && stats,
              "clip + SGD step: bad arguments");
    lsn::SgdArgs a;
    a.t = tensors_dev, a.n = n_tensors, a.chunks = total_chunks, a.max_norm = max_norm;
    for (int i = 0; i < n_groups; ++i) a.g[i] = groups[i];
    a.partial = reinterpret_cast<double *>(workspace);
    a.stats = stats;
    const int blocks = (int)(total_chunks < 2048 ? total_chunks : 2048);
    if (max_norm > 0.f) {
        hipLaunchKernelGGL(lsn::sgd_sqnorm_kernel, dim3(blocks), dim3(256), 0, stream, a);
        hipLaunchKernelGGL(lsn::sgd_coef_kernel, dim3(1), dim3(256), 0, stream, a, blocks);
    }
    hipLaunchKernelGGL(lsn::sgd_step_kernel, dim3(blocks), dim3(256), 0, stream, a);
    LSN_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
