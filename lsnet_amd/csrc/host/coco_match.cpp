// Greedy detection <-> ground-truth matching of COCO evaluation for one (image, category, area range), all IoU
// thresholds at once (reference: cocoapi/pycocotools/pycocotools/cocoeval.py:212-247, a triple Python loop there).
#include "../../../include/lsnet_host.h"

#include <algorithm>

extern "C" void lsn_coco_match(const double *ious, size_t D, size_t G, const uint8_t *gt_ignore,
                               const uint8_t *gt_crowd, const double *thrs, size_t T, int64_t *dt_match,
                               int64_t *gt_match)
{
    std::fill(dt_match, dt_match + T * D, static_cast<int64_t>(-1));
    std::fill(gt_match, gt_match + T * G, static_cast<int64_t>(-1));
    for (size_t t = 0; t < T; ++t) {
        int64_t *gm = gt_match + t * G;
        for (size_t d = 0; d < D; ++d) {           // detections arrive by descending score
            double best = std::min(thrs[t], 1 - 1e-10);
            int64_t m = -1;
            for (size_t g = 0; g < G; ++g) {       // ground truths arrive with the ignored ones last
                if (gm[g] >= 0 && !gt_crowd[g]) continue;                 // taken (a crowd region can be re-used)
                if (m > -1 && !gt_ignore[m] && gt_ignore[g]) break;       // a real match beats any ignored region
                const double v = ious[d * G + g];
                if (v < best) continue;
                best = v;                                                 // ties go to the later ground truth
                m = static_cast<int64_t>(g);
            }
            if (m < 0) continue;
            dt_match[t * D + d] = m;
            gm[m] = static_cast<int64_t>(d);
        }
    }
}
