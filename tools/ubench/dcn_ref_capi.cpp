// C entry points over dcn_ref.h for tests/test_ubench_ref.py (g++ -O2 -shared -fPIC; no GPU involved): every output and
// gradient of ONE level, dense, so that the harness's host reference can be compared with the oracle element by element.
#include "dcn_ref.h"

using namespace dcnref;

extern "C" {

// x (B,H,W,C), off (B,Ho,Wo,och), gout (B,Ho,Wo,Co), w (Co,3,3,C), bias (Co) or NULL -- all channels-last float32.
// out (B,Ho,Wo,Co); gx (B,H,W,C); goff (B,Ho,Wo,och) [offsets, then logit gradients when och == 27]; gw (Co,3,3,C); gb (Co)
void dcnref_all(int B, int H, int W, int Ho, int Wo, int och, float sh, float sw, int C, int Co, const float *x,
                const float *off, const float *gout, const float *w, const float *bias, double *out, double *gx,
                double *goff, double *gw, double *gb)
{
    Lv L = {B, H, W, Ho, Wo, och, sh, sw, x, off, gout};
    for (int b = 0; b < B; ++b)
        for (int ho = 0; ho < Ho; ++ho)
            for (int wo = 0; wo < Wo; ++wo) {
                const size_t px = (size_t)(b * Ho + ho) * Wo + wo;
                for (int co = 0; co < Co; ++co) out[px * Co + co] = forward_at(L, w, bias, C, Co, b, ho, wo, co);
                for (int k = 0; k < K; ++k) {
                    double gy, gxx, gm;
                    goff_at(L, w, C, Co, b, ho, wo, k, &gy, &gxx, &gm);
                    goff[px * och + 2 * k] = gy, goff[px * och + 2 * k + 1] = gxx;
                    if (och == 3 * K) goff[px * och + 2 * K + k] = gm;
                }
            }
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int xx = 0; xx < W; ++xx)
                for (int c = 0; c < C; ++c) gx[((size_t)(b * H + y) * W + xx) * C + c] = gx_at(L, w, C, Co, b, y, xx, c);
    for (int co = 0; co < Co; ++co)
        for (int k = 0; k < K; ++k)
            for (int c = 0; c < C; ++c) {
                double s, sb;
                gw_at(L, C, Co, co, k, c, &s, &sb);
                gw[((size_t)co * K + k) * C + c] = s;
                gb[co] = sb;
            }
}
}
