/*
 * lsnet_oracle.c -- CPU restatement of the reference's native ops on the LSNet hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under lsnet_amd/ may import, link or call this file.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker
 * (or the timed CPU baseline), never as the product path.
 *
 * The reference has no CPU implementation of these ops (they raise NotImplementedError on CPU
 * tensors, mmdet/ops/dcn/deform_conv.py:46-47,136-137,221-222), so this file restates the CUDA
 * algorithm of /root/reference/code/mmdet/ops/dcn/src/cuda/deform_conv_cuda_kernel.cu (im2col /
 * col2im / col2im_coord kernels) and the host GEMM layout of deform_conv_cuda.cpp, plus
 * sigmoid_focal_loss_cuda.cu and nms_cpu.cpp.  Each function cites the lines it follows.
 *
 * Parity pinning: the reference holds no golden vectors for DCN / pyramid DCN / focal loss
 * (SURVEY.md section 0.3); they are pinned by (a) analytic identities checked in
 * tests/test_oracle.py (zero offsets + unit mask == dense convolution; pyramid with scale 1 ==
 * DCNv1; finite differences for every gradient) and (b) the reference's own Python driving this
 * oracle end-to-end (oracle/ref_harness).  NMS is pinned by the reference's known-answer test
 * (tests/test_ops/test_nms.py:18-24) and by oracle/_ref (the reference nms_cpu.cpp compiled).
 *
 * All tensors are contiguous NCHW float32, exactly what the reference extension receives.
 * Arithmetic is plain IEEE float (build with -ffp-contract=off), accumulation in float.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * Bilinear helpers.  deform_conv_cuda_kernel.cu:84-115 (value), :117-143 (gradient weight),
 * :145-188 (coordinate weight); the dmcn_* twins at :745-845 are textually the same maths.
 * ---------------------------------------------------------------------------------------- */
static float bilinear_at(const float *im, int data_width, int height, int width, float h, float w)
{
    int h_low = (int)floorf(h), w_low = (int)floorf(w);
    int h_high = h_low + 1, w_high = w_low + 1;
    float lh = h - (float)h_low, lw = w - (float)w_low;
    float hh = 1.0f - lh, hw = 1.0f - lw;
    float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
    if (h_low >= 0 && w_low >= 0) v1 = im[h_low * data_width + w_low];
    if (h_low >= 0 && w_high <= width - 1) v2 = im[h_low * data_width + w_high];
    if (h_high <= height - 1 && w_low >= 0) v3 = im[h_high * data_width + w_low];
    if (h_high <= height - 1 && w_high <= width - 1) v4 = im[h_high * data_width + w_high];
    float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

static float coord_weight(float ah, float aw, int height, int width, const float *im,
                          int data_width, int bp_dir)
{
    if (ah <= -1.f || ah >= (float)height || aw <= -1.f || aw >= (float)width) return 0.f;
    int hl = (int)floorf(ah), wl = (int)floorf(aw);
    int hh = hl + 1, wh = wl + 1;
    float weight = 0.f;
    if (bp_dir == 0) {
        if (hl >= 0 && wl >= 0) weight += -1.f * ((float)wl + 1.f - aw) * im[hl * data_width + wl];
        if (hl >= 0 && wh <= width - 1) weight += -1.f * (aw - (float)wl) * im[hl * data_width + wh];
        if (hh <= height - 1 && wl >= 0) weight += ((float)wl + 1.f - aw) * im[hh * data_width + wl];
        if (hh <= height - 1 && wh <= width - 1) weight += (aw - (float)wl) * im[hh * data_width + wh];
    } else {
        if (hl >= 0 && wl >= 0) weight += -1.f * ((float)hl + 1.f - ah) * im[hl * data_width + wl];
        if (hl >= 0 && wh <= width - 1) weight += ((float)hl + 1.f - ah) * im[hl * data_width + wh];
        if (hh <= height - 1 && wl >= 0) weight += -1.f * (ah - (float)hl) * im[hh * data_width + wl];
        if (hh <= height - 1 && wh <= width - 1) weight += (ah - (float)hl) * im[hh * data_width + wh];
    }
    return weight;
}

static float grad_weight(float ah, float aw, int h, int w, int height, int width)
{
    if (ah <= -1.f || ah >= (float)height || aw <= -1.f || aw >= (float)width) return 0.f;
    int hl = (int)floorf(ah), wl = (int)floorf(aw);
    int hh = hl + 1, wh = wl + 1;
    float weight = 0.f;
    if (h == hl && w == wl) weight = ((float)h + 1.f - ah) * ((float)w + 1.f - aw);
    if (h == hl && w == wh) weight = ((float)h + 1.f - ah) * (aw + 1.f - (float)w);
    if (h == hh && w == wl) weight = (ah + 1.f - (float)h) * ((float)w + 1.f - aw);
    if (h == hh && w == wh) weight = (ah + 1.f - (float)h) * (aw + 1.f - (float)w);
    return weight;
}

/* Geometry of one deformable convolution call.  `H,W` is the sampled (source) map; `Ho,Wo` the
 * output grid, which for the pyramid variant is the OFFSET grid (deform_conv.py:215-217).
 * scale_h/scale_w are 1 for DCNv1/v2; for the pyramid op they multiply the regular grid position
 * before the offset is added (deform_conv_cuda_kernel.cu:281-282). */
typedef struct {
    int B, C, H, W;       /* input                                   */
    int Co, Ho, Wo;       /* output                                  */
    int kh, kw, sh, sw, ph, pw, dh, dw;
    int groups, dg;       /* conv groups, deformable groups          */
    float scale_h, scale_w;
} orc_geom;

static inline void sample_pos(const orc_geom *g, const float *off_b, int dgi, int i, int j, int ho,
                              int wo, float *py, float *px)
{
    const int K = g->kh * g->kw, k = i * g->kw + j, HW = g->Ho * g->Wo;
    const float *o = off_b + (size_t)dgi * 2 * K * HW;
    float oh = o[(size_t)(2 * k) * HW + ho * g->Wo + wo];
    float ow = o[(size_t)(2 * k + 1) * HW + ho * g->Wo + wo];
    int h_in = ho * g->sh - g->ph, w_in = wo * g->sw - g->pw;
    /* kernel.cu:227-228 (v1), :281-282 (pyramid), :892-893 (v2); scale == 1 reduces to v1/v2 */
    *py = (float)(h_in + i * g->dh) * g->scale_h + oh;
    *px = (float)(w_in + j * g->dw) * g->scale_w + ow;
}

/* im2col for ONE image: col[(c*K + k), ho*Wo + wo].  kernel.cu:191-243 / 246-297 / 848-910.
 * mask == NULL  <=>  DCNv1 / pyramid. */
static void im2col_one(const orc_geom *g, const float *x_b, const float *off_b, const float *mask_b,
                       float *col)
{
    const int K = g->kh * g->kw, HW = g->Ho * g->Wo, cpg = g->C / g->dg;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < g->C; ++c) {
        const int dgi = c / cpg;
        const float *im = x_b + (size_t)c * g->H * g->W;
        for (int i = 0; i < g->kh; ++i)
            for (int j = 0; j < g->kw; ++j) {
                const int k = i * g->kw + j;
                float *dst = col + ((size_t)c * K + k) * HW;
                for (int ho = 0; ho < g->Ho; ++ho)
                    for (int wo = 0; wo < g->Wo; ++wo) {
                        float py, px, val = 0.f;
                        sample_pos(g, off_b, dgi, i, j, ho, wo, &py, &px);
                        if (py > -1.f && px > -1.f && py < (float)g->H && px < (float)g->W)
                            val = bilinear_at(im, g->W, g->H, g->W, py, px);
                        if (mask_b)
                            val *= mask_b[((size_t)dgi * K + k) * HW + ho * g->Wo + wo];
                        dst[ho * g->Wo + wo] = val;
                    }
            }
    }
}

/* out[M,N] (+)= A[M,Kd] * Bm[Kd,N]   (row-major), float accumulation */
static void gemm_nn(int M, int N, int Kd, const float *A, const float *Bm, float *out, int accumulate)
{
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        float *o = out + (size_t)m * N;
        if (!accumulate) memset(o, 0, sizeof(float) * N);
        for (int k = 0; k < Kd; ++k) {
            const float a = A[(size_t)m * Kd + k];
            const float *b = Bm + (size_t)k * N;
            for (int n = 0; n < N; ++n) o[n] += a * b[n];
        }
    }
}

/* out[M,N] (+)= A[Kd,M]^T * Bm[Kd,N] */
static void gemm_tn(int M, int N, int Kd, const float *A, const float *Bm, float *out, int accumulate)
{
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        float *o = out + (size_t)m * N;
        if (!accumulate) memset(o, 0, sizeof(float) * N);
        for (int k = 0; k < Kd; ++k) {
            const float a = A[(size_t)k * M + m];
            const float *b = Bm + (size_t)k * N;
            for (int n = 0; n < N; ++n) o[n] += a * b[n];
        }
    }
}

/* out[M,N] += A[M,Kd] * Bm[N,Kd]^T */
static void gemm_nt_acc(int M, int N, int Kd, const float *A, const float *Bm, float *out)
{
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            const float *a = A + (size_t)m * Kd, *b = Bm + (size_t)n * Kd;
            float s = 0.f;
            for (int k = 0; k < Kd; ++k) s += a[k] * b[k];
            out[(size_t)m * N + n] += s;
        }
}

/* ------------------------------------------------------------------------------------------
 * Forward.  Host flow of deform_conv_cuda.cpp:613-692 (v2: per image im2col -> addmm_ -> +bias),
 * :274-372 (v1) and :811-919 (pyramid); the per-group GEMM is out[g] = W[g] (Co/g x C/g*K) @ col[g].
 * ---------------------------------------------------------------------------------------- */
int orc_deform_conv_forward(const float *x, const float *weight, const float *bias, const float *offset,
                            const float *mask, float *out, int B, int C, int H, int W, int Co, int Ho,
                            int Wo, int kh, int kw, int stride, int pad, int dil, int groups, int dg,
                            float scale_h, float scale_w)
{
    orc_geom g = {B, C, H, W, Co, Ho, Wo, kh, kw, stride, stride, pad, pad, dil, dil, groups, dg,
                  scale_h, scale_w};
    const int K = kh * kw, HW = Ho * Wo, Cg = C / groups, Cog = Co / groups;
    if (C % groups || Co % groups || C % dg) return -1;
    float *col = (float *)malloc(sizeof(float) * (size_t)C * K * HW);
    if (!col) return -2;
    for (int b = 0; b < B; ++b) {
        const float *x_b = x + (size_t)b * C * H * W;
        const float *off_b = offset + (size_t)b * dg * 2 * K * HW;
        const float *m_b = mask ? mask + (size_t)b * dg * K * HW : NULL;
        im2col_one(&g, x_b, off_b, m_b, col);
        for (int gi = 0; gi < groups; ++gi)
            gemm_nn(Cog, HW, Cg * K, weight + (size_t)gi * Cog * Cg * K, col + (size_t)gi * Cg * K * HW,
                    out + ((size_t)b * Co + gi * Cog) * HW, 0);
        if (bias)
            for (int co = 0; co < Co; ++co) {
                float *o = out + ((size_t)b * Co + co) * HW;
                for (int p = 0; p < HW; ++p) o[p] += bias[co];
            }
    }
    free(col);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Backward.  deform_conv_cuda.cpp:694-808 (v2), :374-611 (v1), :921-1153 (pyramid):
 *   gcol = W^T @ gout            (:747-748, :455-456, :994-995)
 *   grad_offset / grad_mask      col2im_coord  kernel.cu:487-549 / 552-615 / 973-1044
 *   grad_input                   col2im        kernel.cu:334-389 / 392-448 / 913-970
 *   grad_weight += gout @ col^T  (:783-787, :583-588, :1126-1131), grad_bias = sum gout (:789-793)
 * Every grad buffer is OVERWRITTEN here except that weight/bias grads accumulate over the batch
 * starting from zero (the Python side passes zeros_like, deform_conv.py:157-161).
 * Any of gx/goff/gmask/gw/gb may be NULL to skip it.
 * ---------------------------------------------------------------------------------------- */
int orc_deform_conv_backward(const float *x, const float *weight, const float *offset, const float *mask,
                             const float *gout, float *gx, float *goff, float *gmask, float *gw, float *gb,
                             int B, int C, int H, int W, int Co, int Ho, int Wo, int kh, int kw,
                             int stride, int pad, int dil, int groups, int dg, float scale_h,
                             float scale_w)
{
    orc_geom g = {B, C, H, W, Co, Ho, Wo, kh, kw, stride, stride, pad, pad, dil, dil, groups, dg,
                  scale_h, scale_w};
    const int K = kh * kw, HW = Ho * Wo, Cg = C / groups, Cog = Co / groups, cpg = C / dg;
    if (C % groups || Co % groups || C % dg) return -1;
    float *col = (float *)malloc(sizeof(float) * (size_t)C * K * HW);
    float *gcol = (float *)malloc(sizeof(float) * (size_t)C * K * HW);
    if (!col || !gcol) { free(col); free(gcol); return -2; }
    if (gx) memset(gx, 0, sizeof(float) * (size_t)B * C * H * W);
    if (gw) memset(gw, 0, sizeof(float) * (size_t)Co * Cg * K);
    if (gb) memset(gb, 0, sizeof(float) * (size_t)Co);

    for (int b = 0; b < B; ++b) {
        const float *x_b = x + (size_t)b * C * H * W;
        const float *off_b = offset + (size_t)b * dg * 2 * K * HW;
        const float *m_b = mask ? mask + (size_t)b * dg * K * HW : NULL;
        const float *go_b = gout + (size_t)b * Co * HW;

        for (int gi = 0; gi < groups; ++gi)
            gemm_tn(Cg * K, HW, Cog, weight + (size_t)gi * Cog * Cg * K, go_b + (size_t)gi * Cog * HW,
                    gcol + (size_t)gi * Cg * K * HW, 0);

        /* col2im_coord: one value per (offset channel, ho, wo); loops the channels of its
         * deformable group.  grad_mask is written by the even (dy) offset channel. */
        if (goff || gmask) {
#pragma omp parallel for schedule(static)
            for (int oc = 0; oc < dg * 2 * K; ++oc) {
                const int dgi = oc / (2 * K), offset_c = oc - dgi * 2 * K;
                const int k = offset_c / 2, i = k / kw, j = k % kw, bp_dir = offset_c % 2;
                for (int ho = 0; ho < Ho; ++ho)
                    for (int wo = 0; wo < Wo; ++wo) {
                        float py, px, val = 0.f, mval = 0.f;
                        sample_pos(&g, off_b, dgi, i, j, ho, wo, &py, &px);
                        const int inside = !(py <= -1.f || px <= -1.f || py >= (float)H || px >= (float)W);
                        if (!inside) py = px = -2.f;
                        const float m = m_b ? m_b[((size_t)dgi * K + k) * HW + ho * Wo + wo] : 1.f;
                        for (int cc = 0; cc < cpg; ++cc) {
                            const int c = dgi * cpg + cc;
                            const float *im = x_b + (size_t)c * H * W;
                            const float gc = gcol[((size_t)c * K + k) * HW + ho * Wo + wo];
                            if (inside && gmask)
                                mval += gc * bilinear_at(im, W, H, W, py, px);
                            val += coord_weight(py, px, H, W, im, W, bp_dir) * gc * m;
                        }
                        if (goff) goff[((size_t)b * dg * 2 * K + oc) * HW + ho * Wo + wo] = val;
                        if (gmask && m_b && bp_dir == 0)
                            gmask[((size_t)(b * dg + dgi) * K + k) * HW + ho * Wo + wo] = mval;
                    }
            }
        }

        /* col2im: scatter with the bilinear weights (the reference scans a 5x5 window around the
         * truncated position, kernel.cu:370-386; the set of taps it finds is the <=4 in-range
         * bilinear corners). Parallel over channels so no two threads touch the same plane. */
        if (gx) {
#pragma omp parallel for schedule(static)
            for (int c = 0; c < C; ++c) {
                const int dgi = c / cpg;
                float *gim = gx + ((size_t)b * C + c) * H * W;
                for (int k = 0; k < K; ++k) {
                    const int i = k / kw, j = k % kw;
                    for (int ho = 0; ho < Ho; ++ho)
                        for (int wo = 0; wo < Wo; ++wo) {
                            float py, px;
                            sample_pos(&g, off_b, dgi, i, j, ho, wo, &py, &px);
                            const float m = m_b ? m_b[((size_t)dgi * K + k) * HW + ho * Wo + wo] : 1.f;
                            const float top = gcol[((size_t)c * K + k) * HW + ho * Wo + wo] * m;
                            const int ch = (int)py, cw = (int)px;
                            for (int dy = -2; dy <= 2; ++dy)
                                for (int dx = -2; dx <= 2; ++dx) {
                                    const int yy = ch + dy, xx = cw + dx;
                                    if (yy >= 0 && yy < H && xx >= 0 && xx < W &&
                                        fabsf(py - (float)yy) < 1.f && fabsf(px - (float)xx) < 1.f)
                                        gim[yy * W + xx] += grad_weight(py, px, yy, xx, H, W) * top;
                                }
                        }
                }
            }
        }

        if (gw || 0) {
            im2col_one(&g, x_b, off_b, m_b, col);
            for (int gi = 0; gi < groups; ++gi)
                gemm_nt_acc(Cog, Cg * K, HW, go_b + (size_t)gi * Cog * HW, col + (size_t)gi * Cg * K * HW,
                            gw + (size_t)gi * Cog * Cg * K);
        }
        if (gb)
            for (int co = 0; co < Co; ++co) {
                const float *o = go_b + (size_t)co * HW;
                float s = 0.f;
                for (int p = 0; p < HW; ++p) s += o[p];
                gb[co] += s;
            }
    }
    free(col);
    free(gcol);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Sigmoid focal loss.  sigmoid_focal_loss_cuda.cu:24-59 (forward), :62-97 (backward).
 * logits (N,C) float, targets (N) int64 in [0,C]; C (= num_classes) is the background label and
 * makes every class a negative (cu:36-37: c1 = (t == d), c2 = (t >= 0 & t != d)).
 * The kernel mixes double literals with float data (`1. / (1. + expf(-x))`, `1. - p`, ...), so
 * the intermediate arithmetic below is carried in double and rounded to float exactly where the
 * CUDA source assigns to scalar_t or calls a float intrinsic.
 * ---------------------------------------------------------------------------------------- */
#define ORC_FLT_MIN 1.17549435e-38f
static inline float fl_sigmoid(float x) { return (float)(1.0 / (1.0 + (double)expf(-x))); }
/* -x*[x>=0] - log(1 + exp(x - 2x[x>=0]))  ==  log(1 - sigmoid(x)), in the kernel's mixed precision */
static inline double fl_log1mp(float x)
{
    const double ge = (x >= 0.f) ? 1.0 : 0.0;
    const float e = expf((float)((double)x - 2.0 * (double)x * ge));
    return -1.0 * (double)x * ge - (double)logf((float)(1.0 + (double)e));
}

void orc_sigmoid_focal_loss_forward(const float *logits, const int64_t *targets, float *losses, int N,
                                    int C, float gamma, float alpha)
{
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        const int t = (int)targets[n];
        for (int d = 0; d < C; ++d) {
            const float x = logits[(size_t)n * C + d];
            const float c1 = (t == d) ? 1.f : 0.f;
            const float c2 = ((t >= 0) & (t != d)) ? 1.f : 0.f;
            const float zn = (float)(1.0 - (double)alpha), zp = alpha;
            const float p = fl_sigmoid(x);
            const float term1 = powf((float)(1.0 - (double)p), gamma) * logf(fmaxf(p, ORC_FLT_MIN));
            const float term2 = (float)((double)powf(p, gamma) * fl_log1mp(x));
            float l = 0.f;
            l += -c1 * term1 * zp;
            l += -c2 * term2 * zn;
            losses[(size_t)n * C + d] = l;
        }
    }
}

void orc_sigmoid_focal_loss_backward(const float *logits, const int64_t *targets, const float *dloss,
                                     float *dlogits, int N, int C, float gamma, float alpha)
{
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        const int t = (int)targets[n];
        for (int d = 0; d < C; ++d) {
            const float x = logits[(size_t)n * C + d];
            const float c1 = (t == d) ? 1.f : 0.f;
            const float c2 = ((t >= 0) & (t != d)) ? 1.f : 0.f;
            const float zn = (float)(1.0 - (double)alpha), zp = alpha;
            const float p = fl_sigmoid(x);
            const float term1 = (float)((double)powf((float)(1.0 - (double)p), gamma) *
                                        (1.0 - (double)p - (double)(p * gamma * logf(fmaxf(p, ORC_FLT_MIN)))));
            const float term2 = (float)((double)powf(p, gamma) *
                                        (fl_log1mp(x) * (1.0 - (double)p) * (double)gamma - (double)p));
            float v = 0.f;
            v += -c1 * term1 * zp;
            v += -c2 * term2 * zn;
            dlogits[(size_t)n * C + d] = v * dloss[(size_t)n * C + d];
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Greedy NMS.  nms_cpu.cpp:8-66: order = argsort(score, descending) (stable here: ties keep
 * input order, as at::sort on CPU with stable ordering of equal keys is what the reference's
 * test vectors exercise), area = (x2-x1)*(y2-y1), suppress when IoU > thr (strict).
 * dets (n,5) = x1,y1,x2,y2,score.  keep (capacity n) receives indices into the input order,
 * in descending-score order.  Returns the number kept.
 * ---------------------------------------------------------------------------------------- */
typedef struct { float s; int i; } orc_si;
static int cmp_si(const void *a, const void *b)
{
    const orc_si *x = (const orc_si *)a, *y = (const orc_si *)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return x->i - y->i;
}

int orc_nms(const float *dets, int n, float thr, int64_t *keep)
{
    if (n <= 0) return 0;
    orc_si *ord = (orc_si *)malloc(sizeof(orc_si) * n);
    unsigned char *supp = (unsigned char *)calloc(n, 1);
    float *area = (float *)malloc(sizeof(float) * n);
    for (int i = 0; i < n; ++i) {
        ord[i].s = dets[i * 5 + 4];
        ord[i].i = i;
        area[i] = (dets[i * 5 + 2] - dets[i * 5 + 0]) * (dets[i * 5 + 3] - dets[i * 5 + 1]);
    }
    qsort(ord, n, sizeof(orc_si), cmp_si);
    int nk = 0;
    for (int _i = 0; _i < n; ++_i) {
        const int i = ord[_i].i;
        if (supp[i]) continue;
        keep[nk++] = i;
        const float ix1 = dets[i * 5], iy1 = dets[i * 5 + 1], ix2 = dets[i * 5 + 2], iy2 = dets[i * 5 + 3];
        for (int _j = _i + 1; _j < n; ++_j) {
            const int j = ord[_j].i;
            if (supp[j]) continue;
            const float xx1 = fmaxf(ix1, dets[j * 5]), yy1 = fmaxf(iy1, dets[j * 5 + 1]);
            const float xx2 = fminf(ix2, dets[j * 5 + 2]), yy2 = fminf(iy2, dets[j * 5 + 3]);
            const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
            const float inter = w * h;
            const float ovr = inter / (area[i] + area[j] - inter);
            if (ovr > thr) supp[j] = 1;
        }
    }
    free(ord);
    free(supp);
    free(area);
    return nk;
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
