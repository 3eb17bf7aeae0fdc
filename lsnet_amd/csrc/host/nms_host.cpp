// liblsnet_host.so -- the two host-only members of the reference's `nms_ext` module (mmdet/ops/nms/src/nms_ext.cpp:29-43
// -> src/cpu/nms_cpu.cpp:63-258): soft NMS and NMS matching.  (The hard NMS of the hot path is the device kernel
// lsn_nms in liblsnet_hip.so.)  float32 boxes [x1, y1, x2, y2, score], the reference's arithmetic and visiting order:
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

}  // namespace

extern "C" {

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
