// Host evaluation (double precision) of single elements of a 3x3 / stride 1 / pad 1 deformable convolution (DCNv1, DCNv2
// with mask logits behind the offsets, pyramid form with per-level scales) on channels-last tensors: what the torch-free
// harness tools/ubench/dcn_step.hip checks the library against.  Semantics of deform_conv_cuda_kernel.cu:84-188 (bilinear
// value / coordinate weights), 227-290 (sampling positions), 392-448 (column -> image), 913-1044 (modulated forms),
// restated here because tools may not link oracle/; tests/test_ubench_ref.py pins THIS file against the oracle on CPU.
#pragma once
#include <cmath>
#include <cstddef>

namespace dcnref {

constexpr int KH = 3, K = 9, PAD = 1;

struct Lv {           // one (source map, offset field, output grid) triple, channels-last
    int B, H, W;      // source map x: (B, H, W, C)
    int Ho, Wo;       // output grid = offset grid
    int och;          // channels of the offset tensor: 18, or 27 with the mask LOGITS behind the offsets (fused)
    float sh, sw;     // grid scales of the pyramid form (1 otherwise)
    const float *x, *off, *gout;   // gout: (B, Ho, Wo, Co), may be NULL for forward-only use
    bool fused() const { return och == 3 * K; }
};

struct Pos {
    bool in;
    int y0, x0;
    double ly, lx;
    bool v[4];   // corner validity: (y0,x0) (y0,x1) (y1,x0) (y1,x1)
};

inline Pos position(const Lv &L, int b, int ho, int wo, int k)
{
    const int i = k / KH, j = k % KH;
    const float *o = L.off + ((size_t)(b * L.Ho + ho) * L.Wo + wo) * L.och;
    // float arithmetic as in the kernels (deform_conv_cuda_kernel.cu:281-282): base * scale, then + offset
    const float by = (float)(ho - PAD + i) * L.sh, bx = (float)(wo - PAD + j) * L.sw;
    const float py = by + o[2 * k], px = bx + o[2 * k + 1];
    Pos p = {};
    p.in = py > -1.f && px > -1.f && py < (float)L.H && px < (float)L.W;
    if (!p.in) return p;
    const float fy = floorf(py), fx = floorf(px);
    p.y0 = (int)fy, p.x0 = (int)fx, p.ly = (double)(py - fy), p.lx = (double)(px - fx);
    p.v[0] = p.y0 >= 0 && p.x0 >= 0, p.v[1] = p.y0 >= 0 && p.x0 + 1 <= L.W - 1;
    p.v[2] = p.y0 + 1 <= L.H - 1 && p.x0 >= 0, p.v[3] = p.y0 + 1 <= L.H - 1 && p.x0 + 1 <= L.W - 1;
    return p;
}

inline double mask_of(const Lv &L, int b, int ho, int wo, int k)   // sigmoid of the logit; 1 without a mask
{
    if (!L.fused()) return 1.0;
    const double m = L.off[((size_t)(b * L.Ho + ho) * L.Wo + wo) * L.och + 2 * K + k];
    return 1.0 / (1.0 + exp(-m));
}

inline double xat(const Lv &L, int C, int b, int y, int x, int c) { return L.x[((size_t)(b * L.H + y) * L.W + x) * C + c]; }

inline void corner_w(const Pos &p, double w[4])
{
    w[0] = (1 - p.ly) * (1 - p.lx), w[1] = (1 - p.ly) * p.lx, w[2] = p.ly * (1 - p.lx), w[3] = p.ly * p.lx;
}

inline double sample(const Lv &L, int C, const Pos &p, int b, int c)   // the bilinear value (without the mask)
{
    if (!p.in) return 0.0;
    double w[4];
    corner_w(p, w);
    double v = 0;
    for (int q = 0; q < 4; ++q)
        if (p.v[q]) v += w[q] * xat(L, C, b, p.y0 + (q >> 1), p.x0 + (q & 1), c);
    return v;
}

// weight: (Co, 3, 3, C) channels-last
inline double forward_at(const Lv &L, const float *w, const float *bias, int C, int Co, int b, int ho, int wo, int co)
{
    (void)Co;
    double s = bias ? bias[co] : 0.0;
    for (int k = 0; k < K; ++k) {
        const Pos p = position(L, b, ho, wo, k);
        if (!p.in) continue;
        const double m = mask_of(L, b, ho, wo, k);
        for (int c = 0; c < C; ++c) s += w[((size_t)co * K + k) * C + c] * m * sample(L, C, p, b, c);
    }
    return s;
}

inline double gcol_at(const Lv &L, const float *w, int C, int Co, int b, int ho, int wo, int k, int c)
{
    const float *g = L.gout + ((size_t)(b * L.Ho + ho) * L.Wo + wo) * Co;
    double s = 0;
    for (int co = 0; co < Co; ++co) s += (double)g[co] * w[((size_t)co * K + k) * C + c];
    return s;
}

// gradients of one sample (b, ho, wo, k): d/d(offset y), d/d(offset x), d/d(mask LOGIT) (0 without a mask)
inline void goff_at(const Lv &L, const float *w, int C, int Co, int b, int ho, int wo, int k, double *gy, double *gx, double *gm)
{
    *gy = *gx = *gm = 0;
    const Pos p = position(L, b, ho, wo, k);
    if (!p.in) return;
    const double m = mask_of(L, b, ho, wo, k);
    double wq[4];
    corner_w(p, wq);
    double sy = 0, sx = 0, sm = 0;
    for (int c = 0; c < C; ++c) {
        const double g = gcol_at(L, w, C, Co, b, ho, wo, k, c);
        double v[4];
        for (int q = 0; q < 4; ++q) v[q] = p.v[q] ? xat(L, C, b, p.y0 + (q >> 1), p.x0 + (q & 1), c) : 0.0;
        sy += g * m * (-(1 - p.lx) * v[0] - p.lx * v[1] + (1 - p.lx) * v[2] + p.lx * v[3]);
        sx += g * m * (-(1 - p.ly) * v[0] + (1 - p.ly) * v[1] - p.ly * v[2] + p.ly * v[3]);
        sm += g * (wq[0] * v[0] + wq[1] * v[1] + wq[2] * v[2] + wq[3] * v[3]);
    }
    *gy = sy, *gx = sx;
    if (L.fused()) *gm = sm * m * (1 - m);
}

// contribution of one level to grad_weight[co][k][c] and to grad_bias[co]
inline void gw_at(const Lv &L, int C, int Co, int co, int k, int c, double *gw, double *gb)
{
    double s = 0, sb = 0;
    for (int b = 0; b < L.B; ++b)
        for (int ho = 0; ho < L.Ho; ++ho)
            for (int wo = 0; wo < L.Wo; ++wo) {
                const double g = L.gout[((size_t)(b * L.Ho + ho) * L.Wo + wo) * Co + co];
                sb += g;
                const Pos p = position(L, b, ho, wo, k);
                if (p.in) s += g * mask_of(L, b, ho, wo, k) * sample(L, C, p, b, c);
            }
    *gw = s, *gb = sb;
}

// contribution of one level to grad_input[b][y][x][c] of its source map
inline double gx_at(const Lv &L, const float *w, int C, int Co, int b, int y, int x, int c)
{
    double s = 0;
    for (int ho = 0; ho < L.Ho; ++ho)
        for (int wo = 0; wo < L.Wo; ++wo)
            for (int k = 0; k < K; ++k) {
                const Pos p = position(L, b, ho, wo, k);
                if (!p.in) continue;
                const int dy = y - p.y0, dx = x - p.x0;
                if (dy < 0 || dy > 1 || dx < 0 || dx > 1 || !p.v[dy * 2 + dx]) continue;
                double wq[4];
                corner_w(p, wq);
                s += wq[dy * 2 + dx] * mask_of(L, b, ho, wo, k) * gcol_at(L, w, C, Co, b, ho, wo, k, c);
            }
    return s;
}

}  // namespace dcnref
