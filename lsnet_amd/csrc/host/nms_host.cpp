// liblsnet_host.so -- the host members of the reference's `nms_ext` module (mmdet/ops/nms/src/nms_ext.cpp:11-43 ->
// src/cpu/nms_cpu.cpp): hard NMS of CPU tensors (:7-71), soft NMS and NMS matching (:73-258).  (The hard NMS of the hot
// path is the device kernel lsn_nms in liblsnet_hip.so; the host one serves CPU tensors, as `nms_cpu` does in the
// reference's dispatcher.)  float32 boxes [x1, y1, x2, y2, score], the reference's arithmetic and visiting order:
// results are index-for-index the reference's.
#include "../../../include/lsnet_host.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace {

struct Det {
    float x1, y1, x2, y2, score, area, index;
};

inline float overlap(const Det &a, const Det &b)
{
    const float w = std::max(0.f, std::min(a.x2, b.x2) - std::max(a.x1, b.x1));
    const float h = std::max(0.f, std::min(a.y2, b.y2) - std::max(a.y1, b.y1));
    const float inter = w * h;
    return inter / (a.area + b.area - inter);
}

// nms_cpu_kernel<scalar_t> (nms_cpu.cpp:7-62): boxes visited in `order` (the caller's descending-score sort, so that ties
// are torch's), a box is dropped when its IoU with an earlier kept box EXCEEDS the threshold; areas without the legacy +1
template <typename T>
size_t nms_hard(const T *dets, const int64_t *order, size_t n, float thr, int64_t *keep)
{
    std::vector<T> area(n);
    for (size_t i = 0; i < n; ++i) area[i] = (dets[5 * i + 2] - dets[5 * i]) * (dets[5 * i + 3] - dets[5 * i + 1]);
    std::vector<unsigned char> gone(n, 0);
    size_t kept = 0;
    for (size_t a = 0; a < n; ++a) {
        const int64_t i = order[a];
        if (gone[i]) continue;
        keep[kept++] = i;
        const T *bi = dets + 5 * i;
        for (size_t b = a + 1; b < n; ++b) {
            const int64_t j = order[b];
            if (gone[j]) continue;
            const T *bj = dets + 5 * j;
            const T w = std::max(static_cast<T>(0), std::min(bi[2], bj[2]) - std::max(bi[0], bj[0]));
            const T h = std::max(static_cast<T>(0), std::min(bi[3], bj[3]) - std::max(bi[1], bj[1]));
            const T inter = w * h;
            if (inter / (area[i] + area[j] - inter) > thr) gone[j] = 1;
        }
    }
    return kept;
}

}  // namespace

extern "C" {

size_t lsn_nms_host_f32(const float *dets, const int64_t *order, size_t n, float iou_thr, int64_t *keep)
{
    return nms_hard<float>(dets, order, n, iou_thr, keep);
}

size_t lsn_nms_host_f64(const double *dets, const int64_t *order, size_t n, float iou_thr, int64_t *keep)
{
    return nms_hard<double>(dets, order, n, iou_thr, keep);
}

// Soft NMS (Bodla et al.): repeatedly move the best remaining box to the front, decay the scores of the rest by their
// overlap with it (method 1: x (1 - iou) above the threshold; 2: x exp(-iou^2 / sigma); else hard), and drop boxes
// whose score falls below min_score by swapping them with the last live box.  out: rows [x1, y1, x2, y2, score, index].
size_t lsn_soft_nms(const float *dets, size_t n, float iou_thr, int method, float sigma, float min_score, float *out)
{
    std::vector<Det> d(n);
    for (size_t i = 0; i < n; ++i) {
        const float *r = dets + 5 * i;
        d[i] = {r[0], r[1], r[2], r[3], r[4], (r[2] - r[0]) * (r[3] - r[1]), static_cast<float>(i)};
    }
    size_t live = n;
    for (size_t i = 0; i < live; ++i) {
        size_t best = i;
        for (size_t p = i + 1; p < live; ++p)
            if (d[best].score < d[p].score) best = p;          // first of equal scores wins
        std::swap(d[i], d[best]);
        const Det top = d[i];
        for (size_t p = i + 1; p < live; ++p) {
            const float iou = overlap(top, d[p]);
            float weight = 1.f;
            if (method == 1) {
                if (iou > iou_thr) weight = 1 - iou;
            } else if (method == 2) {
                weight = std::exp(-(iou * iou) / sigma);
            } else {
                weight = iou > iou_thr ? 0.f : 1.f;
            }
            d[p].score = weight * d[p].score;
            if (d[p].score < min_score) {                       // discard: the last live box takes this slot
                d[p] = d[live - 1];
                --live;
                --p;
            }
        }
    }
    for (size_t i = 0; i < live; ++i) {
        float *o = out + 6 * i;
        o[0] = d[i].x1; o[1] = d[i].y1; o[2] = d[i].x2; o[3] = d[i].y2; o[4] = d[i].score; o[5] = d[i].index;
    }
    return live;
}

// NMS matching: boxes visited in `order` (descending score, supplied by the caller so that ties are his); every kept
// box opens a group and takes the not-yet-suppressed boxes with IoU >= thr into it.  flat: the groups back to back
// (keeper first, members in visiting order); group_start[g] .. group_start[g + 1] delimits group g.  Returns #groups.
size_t lsn_nms_match(const float *dets, const int64_t *order, size_t n, float iou_thr, int64_t *flat, int64_t *group_start)
{
    std::vector<Det> d(n);
    for (size_t i = 0; i < n; ++i) {
        const float *r = dets + 5 * i;
        d[i] = {r[0], r[1], r[2], r[3], r[4], (r[2] - r[0]) * (r[3] - r[1]), static_cast<float>(i)};
    }
    std::vector<unsigned char> gone(n, 0);
    size_t groups = 0, w = 0;
    for (size_t a = 0; a < n; ++a) {
        const int64_t i = order[a];
        if (gone[i]) continue;
        group_start[groups++] = static_cast<int64_t>(w);
        flat[w++] = i;
        for (size_t b = a + 1; b < n; ++b) {
            const int64_t j = order[b];
            if (gone[j]) continue;
            if (overlap(d[i], d[j]) >= iou_thr) {
                gone[j] = 1;
                flat[w++] = j;
            }
        }
    }
    group_start[groups] = static_cast<int64_t>(w);
    return groups;
}

}  // extern "C"
