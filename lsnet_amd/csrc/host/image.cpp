// liblsnet_host.so -- image preparation of the data pipeline (C ABI in include/lsnet_host.h): bilinear resize and
// normalisation of HxWxC interleaved images.  The reference takes these from OpenCV through mmcv
// (mmcv/image/geometric.py:26-56 imresize -> cv2.resize INTER_LINEAR, photometric.py:8-41 imnormalize); the
// arithmetic below is the one lsnet_amd/data/geometry.py documents (half-pixel centres, 11-bit fixed-point weights and
// two-pass rounding for 8-bit images; float32 for float images), so both implementations return identical arrays.
#include "../../../include/lsnet_host.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

struct Taps {
    std::vector<int> lo, hi;
    std::vector<float> frac;
};

// source index pair and weight of the right / lower tap for every destination index
Taps linear_taps(int dst, int src)
{
    Taps t;
    t.lo.resize(dst); t.hi.resize(dst); t.frac.resize(dst);
    const double scale = 1.0 / (static_cast<double>(dst) / src);
    for (int d = 0; d < dst; ++d) {
        float f = static_cast<float>((d + 0.5) * scale - 0.5);
        int s = static_cast<int>(std::floor(f));
        f -= static_cast<float>(s);
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= src - 1) { f = 0.f; s = src - 1; }
        t.lo[d] = s;
        t.hi[d] = s + 1 < src ? s + 1 : src - 1;
        t.frac[d] = f;
    }
    return t;
}

template <typename T>
void normalize(const T *src, size_t pixels, int c, const float *mean, const float *inv_std, bool reverse, float *dst)
{
    if (c == 3) {                                   // the colour-image case, unrolled so that it vectorises
        const float m0 = mean[0], m1 = mean[1], m2 = mean[2], s0 = inv_std[0], s1 = inv_std[1], s2 = inv_std[2];
        const int i0 = reverse ? 2 : 0, i2 = reverse ? 0 : 2;
        for (size_t p = 0; p < pixels; ++p) {
            const T *q = src + 3 * p;
            float *o = dst + 3 * p;
            o[0] = (static_cast<float>(q[i0]) - m0) * s0;
            o[1] = (static_cast<float>(q[1]) - m1) * s1;
            o[2] = (static_cast<float>(q[i2]) - m2) * s2;
        }
        return;
    }
    for (size_t p = 0; p < pixels; ++p)
        for (int k = 0; k < c; ++k)
            dst[p * c + k] = (static_cast<float>(src[p * c + (reverse ? c - 1 - k : k)]) - mean[k]) * inv_std[k];
}

}  // namespace

extern "C" {

int lsn_image_resize_bilinear_u8(const uint8_t *src, int sh, int sw, int c, uint8_t *dst, int dh, int dw)
{
    if (sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || c <= 0) return 1;
    const Taps tx = linear_taps(dw, sw), ty = linear_taps(dh, sh);
    std::vector<int> ax0(dw), ax1(dw);
    for (int x = 0; x < dw; ++x) {
        ax1[x] = static_cast<int>(std::nearbyintf(tx.frac[x] * 2048.f));
        ax0[x] = static_cast<int>(std::nearbyintf((1.f - tx.frac[x]) * 2048.f));
    }
    const size_t row_len = static_cast<size_t>(dw) * c;
    std::vector<int> bufa(row_len), bufb(row_len);     // horizontally interpolated source rows (scaled by 2^11)
    int *row[2] = {bufa.data(), bufb.data()};
    int have[2] = {-1, -1};
    auto fill = [&](int slot, int y) {
        const uint8_t *s = src + static_cast<size_t>(y) * sw * c;
        int *o = row[slot];
        for (int x = 0; x < dw; ++x) {
            const uint8_t *p0 = s + static_cast<size_t>(tx.lo[x]) * c, *p1 = s + static_cast<size_t>(tx.hi[x]) * c;
            for (int k = 0; k < c; ++k) o[x * c + k] = p0[k] * ax0[x] + p1[k] * ax1[x];
        }
        have[slot] = y;
    };
    for (int y = 0; y < dh; ++y) {
        const int y0 = ty.lo[y], y1 = ty.hi[y];
        if (have[1] == y0) { std::swap(row[0], row[1]); std::swap(have[0], have[1]); }     // the window slides down
        if (have[0] != y0) fill(0, y0);
        if (have[1] != y1) { if (y1 == y0) { std::memcpy(row[1], row[0], row_len * sizeof(int)); have[1] = y1; } else fill(1, y1); }
        const int ay1 = static_cast<int>(std::nearbyintf(ty.frac[y] * 2048.f));
        const int ay0 = static_cast<int>(std::nearbyintf((1.f - ty.frac[y]) * 2048.f));
        uint8_t *o = dst + static_cast<size_t>(y) * row_len;
        const int *r0 = row[0], *r1 = row[1];
        for (size_t i = 0; i < row_len; ++i) {
            int v = (((ay0 * (r0[i] >> 4)) >> 16) + ((ay1 * (r1[i] >> 4)) >> 16) + 2) >> 2;
            o[i] = static_cast<uint8_t>(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
    return 0;
}

int lsn_image_resize_bilinear_f32(const float *src, int sh, int sw, int c, float *dst, int dh, int dw)
{
    if (sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || c <= 0) return 1;
    const Taps tx = linear_taps(dw, sw), ty = linear_taps(dh, sh);
    const size_t row_len = static_cast<size_t>(dw) * c;
    std::vector<float> bufa(row_len), bufb(row_len);
    float *row[2] = {bufa.data(), bufb.data()};
    int have[2] = {-1, -1};
    auto fill = [&](int slot, int y) {
        const float *s = src + static_cast<size_t>(y) * sw * c;
        float *o = row[slot];
        for (int x = 0; x < dw; ++x) {
            const float *p0 = s + static_cast<size_t>(tx.lo[x]) * c, *p1 = s + static_cast<size_t>(tx.hi[x]) * c;
            const float w1 = tx.frac[x], w0 = 1.f - w1;
            for (int k = 0; k < c; ++k) o[x * c + k] = p0[k] * w0 + p1[k] * w1;
        }
        have[slot] = y;
    };
    for (int y = 0; y < dh; ++y) {
        const int y0 = ty.lo[y], y1 = ty.hi[y];
        if (have[1] == y0) { std::swap(row[0], row[1]); std::swap(have[0], have[1]); }
        if (have[0] != y0) fill(0, y0);
        if (have[1] != y1) { if (y1 == y0) { std::memcpy(row[1], row[0], row_len * sizeof(float)); have[1] = y1; } else fill(1, y1); }
        const float w1 = ty.frac[y], w0 = 1.f - w1;
        float *o = dst + static_cast<size_t>(y) * row_len;
        const float *r0 = row[0], *r1 = row[1];
        for (size_t i = 0; i < row_len; ++i) o[i] = r0[i] * w0 + r1[i] * w1;
    }
    return 0;
}

// dst[p, k] = (src[p, reverse ? c-1-k : k] - mean[k]) * inv_std[k]   (float32 arithmetic, no fused multiply-add)
void lsn_image_normalize_u8(const uint8_t *src, size_t pixels, int c, const float *mean, const float *inv_std,
                            int reverse_channels, float *dst)
{
    normalize(src, pixels, c, mean, inv_std, reverse_channels != 0, dst);
}

void lsn_image_normalize_f32(const float *src, size_t pixels, int c, const float *mean, const float *inv_std,
                             int reverse_channels, float *dst)
{
    normalize(src, pixels, c, mean, inv_std, reverse_channels != 0, dst);
}

}  // extern "C"
