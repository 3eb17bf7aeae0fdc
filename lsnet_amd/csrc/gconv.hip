// Grouped convolution (ResNeXt 64x4d: 3x3, groups = 64, 4 .. 32 channels per group) for gfx950, exact fp32.
//
// Reference: torch.nn.Conv2d(groups=G) built by mmdet/models/backbones/resnext.py:11-83 (Bottleneck.conv2), i.e.
// ATen's cudnn / MIOpen grouped convolution.  Here: three kernels on NHWC tensors with OHWI weights
//   w[co][i][j][ci], co in [0, Co), ci in [0, CG)  (CG = C / G input channels AND Co / G output channels per group).
// Per group the products form a (pixels x 9 CG) x (9 CG x CG) GEMM with CG = 4 .. 32: far too thin for the matrix
// pipe to pay (a 32x32 MFMA tile would be 1/8 .. 1/64 full), and at 9 CG MACs per loaded input value the layer is
// bound by HBM, not by the 157 TFLOP/s fp32 VALU rate.  So: plain fp32 FMA chains (bit-level "reference arithmetic",
// no split products), one lane per pixel, and every weight a WAVE-UNIFORM operand (scalar loads, SGPR sources):
//   forward / data gradient: a wave owns 64 consecutive pixels and one 32-channel slab (128 bytes per pixel: the
//     cache line the neighbouring groups of the slab share); it walks the slab's groups one after the other, per
//     group CG accumulators per lane;
//   weight gradient: a lane owns one (co, ci) pair of one group and the 9 taps; the block walks a range of pixels,
//     every load of a wave falls into one or two cache lines (broadcast); partial sums meet in fp32 atomics.
#include "common.h"

namespace lsn {

struct GcArgs {
    const float *x, *w, *bias, *gout;
    float *out, *gw, *gb;
    int B, H, W, Ho, Wo, C, kh, kw, stride, pad, dil, groups, relu;
    int P;        // pixels of the tensor the lanes walk (forward: B Ho Wo, data gradient: B H W)
    int psplit;   // weight gradient: pixels per block
};

constexpr int GC_SLAB = 32;   // channels a wave covers per pixel: one 128-byte line

// BWD = false: out[p][g CG + co]  = bias + sum_{i,j,ci} x[pix(p, i, j)][g CG + ci] w[g CG + co][i][j][ci]
// BWD = true : gx[q][g CG + ci]   =        sum_{i,j,co} gout[pix'(q, i, j)][g CG + co] w[g CG + co][i][j][ci]
//              (pix': the output pixels whose tap (i, j) reads input pixel q -- (y + pad - i dil) divisible by stride)
template <int CG, bool BWD>
__global__ __launch_bounds__(256) void gconv_kernel(const GcArgs a)
{
    constexpr int GS = GC_SLAB / CG;   // groups per slab
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slab = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + wave);
    const int g0 = slab * GS;
    if (g0 >= a.groups) return;
    const int p = blockIdx.x * 64 + lane;
    const bool pok = p < a.P;
    const int K = a.kh * a.kw;
    // the tensor the lanes walk (OW x OH) and the one they read (IW x IH)
    const int OH = BWD ? a.H : a.Ho, OW = BWD ? a.W : a.Wo, IH = BWD ? a.Ho : a.H, IW = BWD ? a.Wo : a.W;
    const int pp = pok ? p : 0;
    const int b = pp / (OH * OW), rem = pp - b * OH * OW;
    const int oy = rem / OW, ox = rem - oy * OW;
    const float *src = BWD ? a.gout : a.x;
    float *dst = BWD ? a.out : a.out;
    const int Csrc = a.C, Cdst = a.C;   // CG in == CG out: both tensors have groups * CG channels

    const int ng = min(GS, a.groups - g0);
    for (int gi = 0; gi < ng; ++gi) {
        const int g = g0 + gi;
        float acc[CG];
#pragma unroll
        for (int c = 0; c < CG; ++c) acc[c] = (!BWD && a.bias) ? a.bias[g * CG + c] : 0.f;
        for (int i = 0; i < a.kh; ++i)
            for (int j = 0; j < a.kw; ++j) {
                int iy, ix;
                bool ok = pok;
                if (!BWD) {
                    iy = oy * a.stride - a.pad + i * a.dil;
                    ix = ox * a.stride - a.pad + j * a.dil;
                } else {
                    const int ty = oy + a.pad - i * a.dil, tx = ox + a.pad - j * a.dil;
                    iy = ty / a.stride, ix = tx / a.stride;
                    ok = ok && ty >= 0 && tx >= 0 && iy * a.stride == ty && ix * a.stride == tx;
                }
                ok = ok && (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW;
                float v[CG];
                const float *sp = src + ((size_t)(b * IH + (ok ? iy : 0)) * IW + (ok ? ix : 0)) * Csrc + g * CG;
#pragma unroll
                for (int c4 = 0; c4 < CG / 4; ++c4) {
                    const float4 t = *reinterpret_cast<const float4 *>(sp + c4 * 4);
                    v[c4 * 4 + 0] = ok ? t.x : 0.f, v[c4 * 4 + 1] = ok ? t.y : 0.f;
                    v[c4 * 4 + 2] = ok ? t.z : 0.f, v[c4 * 4 + 3] = ok ? t.w : 0.f;
                }
                // weights of (group g, tap): w[(g CG + co) K + tap][ci] -- wave-uniform addresses
                const float *wp = a.w + ((size_t)g * CG * K + (i * a.kw + j)) * CG;
#pragma unroll
                for (int co = 0; co < CG; ++co)
#pragma unroll
                    for (int ci = 0; ci < CG; ++ci) {
                        const float wv = wp[(size_t)co * K * CG + ci];
                        if (!BWD)
                            acc[co] = fmaf(v[ci], wv, acc[co]);
                        else
                            acc[ci] = fmaf(v[co], wv, acc[ci]);
                    }
            }
        if (pok) {
            float *dp = dst + (size_t)p * Cdst + g * CG;
#pragma unroll
            for (int c4 = 0; c4 < CG / 4; ++c4) {
                float4 t = make_float4(acc[c4 * 4], acc[c4 * 4 + 1], acc[c4 * 4 + 2], acc[c4 * 4 + 3]);
                if (!BWD && a.relu) t = make_float4(fmaxf(t.x, 0.f), fmaxf(t.y, 0.f), fmaxf(t.z, 0.f), fmaxf(t.w, 0.f));
                *reinterpret_cast<float4 *>(dp + c4 * 4) = t;
            }
        }
    }
}

// gw[g CG + co][i][j][ci] += sum_p gout[p][g CG + co] x[pix(p, i, j)][g CG + ci];  gb[co] += sum_p gout[p][co]
// thread = pair (co, ci) of one group: q = blockIdx.y * 256 + tid in [0, groups CG CG); block = pixels
// [blockIdx.x psplit, + psplit).  KMAX taps in registers.
template <int CG, int KMAX>
__global__ __launch_bounds__(256) void gconv_wgrad_kernel(const GcArgs a)
{
    const int q = blockIdx.y * 256 + threadIdx.x;
    const int npairs = a.groups * CG * CG;
    const bool qok = q < npairs;
    const int qq = qok ? q : 0;
    const int g = qq / (CG * CG), r = qq - g * CG * CG, co = r / CG, ci = r - co * CG;
    const int K = a.kh * a.kw;
    float acc[KMAX], bsum = 0.f;
#pragma unroll
    for (int t = 0; t < KMAX; ++t) acc[t] = 0.f;
    const int p0 = blockIdx.x * a.psplit, p1 = min(p0 + a.psplit, a.P);
    const int HWo = a.Ho * a.Wo;
    int b = p0 / HWo, rem = p0 - b * HWo;
    int ho = rem / a.Wo, wo = rem - ho * a.Wo;
    const float *gp = a.gout + (size_t)g * CG + co, *xp = a.x + (size_t)g * CG + ci;
    for (int p = p0; p < p1; ++p) {
        const float gv = gp[(size_t)p * a.C];
        bsum += gv;
        const int y0 = ho * a.stride - a.pad, x0 = wo * a.stride - a.pad;
#pragma unroll
        for (int t = 0; t < KMAX; ++t) {
            if (t < K) {
                const int i = t / a.kw, j = t - i * a.kw;
                const int y = y0 + i * a.dil, x = x0 + j * a.dil;
                if ((unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W)   // (uniform over the block)
                    acc[t] = fmaf(gv, xp[((size_t)(b * a.H + y) * a.W + x) * a.C], acc[t]);
            }
        }
        if (++wo == a.Wo) {
            wo = 0;
            if (++ho == a.Ho) ho = 0, ++b;
        }
    }
    if (!qok) return;
#pragma unroll
    for (int t = 0; t < KMAX; ++t)
        if (t < K) atomic_add_f32(a.gw + ((size_t)(g * CG + co) * K + t) * CG + ci, acc[t]);
    if (a.gb && ci == 0) atomic_add_f32(a.gb + g * CG + co, bsum);
}

static int gc_out(int in, int k, int stride, int pad, int dil) { return (in + 2 * pad - (dil * (k - 1) + 1)) / stride + 1; }

static int gc_fill(GcArgs &a, int B, int H, int W, int C, int Co, int kh, int kw, int stride, int pad, int dil, int groups)
{
    LSN_CHECK(B > 0 && H > 0 && W > 0 && C > 0 && Co > 0 && kh > 0 && kw > 0, "grouped conv2d: empty tensor");
    LSN_CHECK(stride > 0 && dil > 0 && pad >= 0 && groups > 0, "grouped conv2d: bad stride / dilation / padding / groups");
    LSN_CHECK(C % groups == 0 && Co % groups == 0, "grouped conv2d: channels (%d -> %d) not divisible by groups = %d", C, Co,
              groups);
    const int cg = C / groups;
    if (cg != Co / groups || !(cg == 4 || cg == 8 || cg == 16 || cg == 32))
        return fail(LSN_ERR_UNSUPPORTED, "grouped conv2d kernels take 4, 8, 16 or 32 channels per group, in = out (got %d -> %d)",
                    cg, Co / groups);
    if (kh * kw > 9) return fail(LSN_ERR_UNSUPPORTED, "grouped conv2d kernels take at most 9 taps, got %d x %d", kh, kw);
    a.B = B, a.H = H, a.W = W, a.C = C, a.kh = kh, a.kw = kw, a.stride = stride, a.pad = pad, a.dil = dil, a.groups = groups;
    a.Ho = gc_out(H, kh, stride, pad, dil), a.Wo = gc_out(W, kw, stride, pad, dil);
    LSN_CHECK(a.Ho > 0 && a.Wo > 0, "grouped conv2d: output size is too small");
    if ((int64_t)B * H * W * C >= ((int64_t)1 << 31) || (int64_t)B * a.Ho * a.Wo * Co >= ((int64_t)1 << 31))
        return fail(LSN_ERR_UNSUPPORTED, "grouped conv2d: tensor too large for 32-bit indexing");
    return 0;
}

template <bool BWD>
static int gc_launch(const GcArgs &a, hipStream_t st)
{
    const int cg = a.C / a.groups;
    dim3 grid(cdiv(a.P, 64), cdiv(cdiv(a.C, GC_SLAB), 4));
    switch (cg) {
    case 4: hipLaunchKernelGGL((gconv_kernel<4, BWD>), grid, dim3(256), 0, st, a); break;
    case 8: hipLaunchKernelGGL((gconv_kernel<8, BWD>), grid, dim3(256), 0, st, a); break;
    case 16: hipLaunchKernelGGL((gconv_kernel<16, BWD>), grid, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((gconv_kernel<32, BWD>), grid, dim3(256), 0, st, a); break;
    }
    LSN_HIP(hipGetLastError());
    return 0;
}

}  // namespace lsn

using namespace lsn;

extern "C" {

int lsn_grouped_conv2d_forward(const float *x, const float *w, const float *bias, float *out, int B, int H, int W, int C,
                               int Co, int kh, int kw, int stride, int pad, int dil, int groups, int relu,
                               lsn_stream_t stream)
{
    LSN_CHECK(x && w && out, "grouped conv2d: NULL tensor");
    GcArgs a = {};
    if (int rc = gc_fill(a, B, H, W, C, Co, kh, kw, stride, pad, dil, groups)) return rc;
    a.x = x, a.w = w, a.bias = bias, a.out = out, a.relu = relu;
    a.P = B * a.Ho * a.Wo;
    return gc_launch<false>(a, static_cast<hipStream_t>(stream));
}

int lsn_grouped_conv2d_backward_data(const float *grad_out, const float *w, float *grad_in, int B, int H, int W, int C,
                                     int Co, int kh, int kw, int stride, int pad, int dil, int groups,
                                     lsn_stream_t stream)
{
    LSN_CHECK(grad_out && w && grad_in, "grouped conv2d backward: NULL tensor");
    GcArgs a = {};
    if (int rc = gc_fill(a, B, H, W, C, Co, kh, kw, stride, pad, dil, groups)) return rc;
    a.gout = grad_out, a.w = w, a.out = grad_in;
    a.P = B * H * W;
    return gc_launch<true>(a, static_cast<hipStream_t>(stream));
}

int lsn_grouped_conv2d_backward_weight(const float *x, const float *grad_out, float *grad_w, float *grad_bias, int B,
                                       int H, int W, int C, int Co, int kh, int kw, int stride, int pad, int dil,
                                       int groups, int accumulate, lsn_stream_t stream)
{
    LSN_CHECK(x && grad_out && grad_w, "grouped conv2d backward-weight: NULL tensor");
    GcArgs a = {};
    if (int rc = gc_fill(a, B, H, W, C, Co, kh, kw, stride, pad, dil, groups)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    a.x = x, a.gout = grad_out, a.gw = grad_w, a.gb = grad_bias;
    a.P = B * a.Ho * a.Wo;
    const int cg = C / groups, K = kh * kw;
    if (!accumulate) {
        LSN_HIP(hipMemsetAsync(grad_w, 0, sizeof(float) * (size_t)Co * K * cg, st));
        if (grad_bias) LSN_HIP(hipMemsetAsync(grad_bias, 0, sizeof(float) * (size_t)Co, st));
    }
    const int ny = cdiv(groups * cg * cg, 256);
    int nsplit = cdiv(4096, ny);   // ~16 blocks per CU in flight: the kernel is load-latency bound
    if (nsplit > cdiv(a.P, 32)) nsplit = cdiv(a.P, 32);
    if (nsplit < 1) nsplit = 1;
    a.psplit = cdiv(a.P, nsplit);
    dim3 grid(cdiv(a.P, a.psplit), ny);
    switch (cg) {
    case 4: hipLaunchKernelGGL((gconv_wgrad_kernel<4, 9>), grid, dim3(256), 0, st, a); break;
    case 8: hipLaunchKernelGGL((gconv_wgrad_kernel<8, 9>), grid, dim3(256), 0, st, a); break;
    case 16: hipLaunchKernelGGL((gconv_wgrad_kernel<16, 9>), grid, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((gconv_wgrad_kernel<32, 9>), grid, dim3(256), 0, st, a); break;
    }
    LSN_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
