// One row of the cross-IOU loss of the bbox task (5 landmarks x [y_up, y_down, x_left, x_right] = 20 components)
// and its gradient with respect to the prediction -- the arithmetic of lsnet_amd/models/losses/cross_iou_loss.py
// (reference: mmdet/models/losses/cross_iou_loss.py:10-33, 61-132), written once for the device kernel
// (csrc/loss.hip) and, compiled by a host compiler, for the CPU check of the hand-derived gradient against autograd
// (tests/test_fused_cross_iou.py).
//
//   target'      : the inactive half of every (neg, pos) pair is alpha x the active half
//   overlap      : sum(min(p, t')) / sum(max(p, t'))
//   box(p)       : [x_left-lm, y_top-lm, x_right-lm, y_bottom-lm] + anchor, each landmark coordinate the signed
//                  combination of its pair (pos if pos > neg else -neg)
//   loss         : 1 - (overlap - (rho^2 / c^2 + v^2 / (1 - overlap + v)))     (centre distance + aspect terms)
// Ties follow ATen: max / min split the gradient 0.5 / 0.5, clamp(min=0) passes the gradient at 0.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define LSN_HD __host__ __device__ __forceinline__
#else
#define LSN_HD static inline
#endif

struct CrossIouRow {
    float loss;        // unweighted
    float grad[20];    // d loss / d pred
};

LSN_HD float ciou_signed(float neg, float pos) { return pos > neg ? pos : -neg; }

// active[c] != 0: component c carries the ground truth; want_grad = 0 skips the gradient
LSN_HD void cross_iou_bbox_row(const float *p, const float *t, const unsigned char *active, const float *anchor,
                               const float *gt, float alpha, float eps, int want_grad, CrossIouRow *out)
{
    float tt[20];
    float smin = 0.f, smax = 0.f;
    for (int j = 0; j < 10; ++j) {
        const float act = active[2 * j] ? t[2 * j] : t[2 * j + 1];
        tt[2 * j] = active[2 * j] ? t[2 * j] : alpha * act;
        tt[2 * j + 1] = active[2 * j + 1] ? t[2 * j + 1] : alpha * act;
    }
    for (int c = 0; c < 20; ++c) {
        smin += fminf(p[c], tt[c]);
        smax += fmaxf(p[c], tt[c]);
    }
    const float ov = smin / smax;
    // landmark k: y = signed(p[4k], p[4k+1]), x = signed(p[4k+2], p[4k+3]); box = [x1, y0, x3, y2] + anchor
    const float bx0 = ciou_signed(p[6], p[7]) + anchor[0], by0 = ciou_signed(p[0], p[1]) + anchor[1];
    const float bx1 = ciou_signed(p[14], p[15]) + anchor[0], by1 = ciou_signed(p[8], p[9]) + anchor[1];
    const float ew_raw = fmaxf(bx1, gt[2]) - fminf(bx0, gt[0]), eh_raw = fmaxf(by1, gt[3]) - fminf(by0, gt[1]);
    const float ew = fmaxf(ew_raw, 0.f), eh = fmaxf(eh_raw, 0.f);
    const float c2 = ew * ew + eh * eh + eps;
    const float w1 = bx1 - bx0, h1 = by1 - by0 + eps;
    const float w2 = gt[2] - gt[0], h2 = gt[3] - gt[1] + eps;
    const float dxs = (gt[0] + gt[2]) - (bx0 + bx1), dys = (gt[1] + gt[3]) - (by0 + by1);
    const float rho2 = dxs * dxs / 4 + dys * dys / 4;
    const float kv = 4.f / (3.14159265358979323846f * 3.14159265358979323846f);
    const float da = atanf(w2 / h2) - atanf(w1 / h1);
    const float v = kv * da * da;
    const float D = 1.f - ov + v;
    out->loss = 1.f - (ov - (rho2 / c2 + v * v / D));
    if (!want_grad) return;

    const float dL_dov = -1.f + v * v / (D * D);
    const float dL_dv = (2.f * v * D - v * v) / (D * D);
    const float dL_drho2 = 1.f / c2, dL_dc2 = -rho2 / (c2 * c2);
    // aspect term: v = kv (A2 - A1)^2, A1 = atan(w1 / h1)
    const float r = w1 / h1, dA1 = 1.f / (1.f + r * r);
    const float dv_dw1 = kv * 2.f * da * (-dA1 / h1), dv_dh1 = kv * 2.f * da * (dA1 * w1 / (h1 * h1));
    // enclosing box: max / min against the ground truth (ties 0.5), clamp at 0 passes the gradient when >= 0
    const float cw = ew_raw >= 0.f ? 1.f : 0.f, ch = eh_raw >= 0.f ? 1.f : 0.f;
    const float mx1 = bx1 > gt[2] ? 1.f : (bx1 == gt[2] ? .5f : 0.f), mn0 = bx0 < gt[0] ? 1.f : (bx0 == gt[0] ? .5f : 0.f);
    const float my1 = by1 > gt[3] ? 1.f : (by1 == gt[3] ? .5f : 0.f), my0 = by0 < gt[1] ? 1.f : (by0 == gt[1] ? .5f : 0.f);
    const float g_bx0 = dL_drho2 * (-dxs / 2) + dL_dc2 * 2.f * ew * cw * (-mn0) + dL_dv * dv_dw1 * (-1.f);
    const float g_bx1 = dL_drho2 * (-dxs / 2) + dL_dc2 * 2.f * ew * cw * mx1 + dL_dv * dv_dw1;
    const float g_by0 = dL_drho2 * (-dys / 2) + dL_dc2 * 2.f * eh * ch * (-my0) + dL_dv * dv_dh1 * (-1.f);
    const float g_by1 = dL_drho2 * (-dys / 2) + dL_dc2 * 2.f * eh * ch * my1 + dL_dv * dv_dh1;
    for (int c = 0; c < 20; ++c) {
        const float dmin = p[c] < tt[c] ? 1.f : (p[c] == tt[c] ? .5f : 0.f);
        const float dmax = p[c] > tt[c] ? 1.f : (p[c] == tt[c] ? .5f : 0.f);
        out->grad[c] = dL_dov * (dmin / smax - smin / (smax * smax) * dmax);
    }
    // box coordinates -> the selected half of their pair: (neg, pos) at (c, c + 1); pos > neg selects +pos else -neg
    const int pair_of[4] = {6, 0, 14, 8};                    // bx0 <- x of lm 1, by0 <- y of lm 0, bx1 <- x of lm 3, by1 <- y of lm 2
    const float g_box[4] = {g_bx0, g_by0, g_bx1, g_by1};
    for (int q = 0; q < 4; ++q) {
        const int c = pair_of[q];
        if (p[c + 1] > p[c]) out->grad[c + 1] += g_box[q]; else out->grad[c] -= g_box[q];
    }
}

// The whole bbox regression stage of one point (lsnet_head.py:1066-1101 via LSHead.loss_levels): prediction in stride
// units -> pixels -> normalised by base_scale * stride; ground-truth extreme points (5 x (x, y)) -> the four-component
// regression target and the active-half mask (lsnet_head.py:402-427: `offset >= 0` selects the positive half; points
// without an object get zero targets); anchor and ground-truth box normalised the same way; then the row loss above.
// out->grad is d loss / d (raw prediction).
LSN_HD void cross_iou_bbox_stage_row(const float *pred_raw, const float *gt_pts, const float *anchor3, const float *gt_box,
                                     int has_object, float base_scale, float alpha, float eps, int want_grad,
                                     CrossIouRow *out)
{
    const float stride = anchor3[2], norm = base_scale * stride;
    float p[20], t[20], an[2], gb[4];
    unsigned char act[20];
    for (int c = 0; c < 20; ++c) p[c] = pred_raw[c] * stride / norm;
    for (int k = 0; k < 5; ++k) {
        const float dx = gt_pts[2 * k] - anchor3[0], dy = gt_pts[2 * k + 1] - anchor3[1];
        const float mx = has_object ? fabsf(dx) : 0.f, my = has_object ? fabsf(dy) : 0.f;
        const bool px = dx >= 0.f, py = dy >= 0.f;
        t[4 * k] = (py ? 0.f : my) / norm;   t[4 * k + 1] = (py ? my : 0.f) / norm;
        t[4 * k + 2] = (px ? 0.f : mx) / norm; t[4 * k + 3] = (px ? mx : 0.f) / norm;
        act[4 * k] = !py; act[4 * k + 1] = py; act[4 * k + 2] = !px; act[4 * k + 3] = px;
    }
    an[0] = anchor3[0] / norm; an[1] = anchor3[1] / norm;
    for (int q = 0; q < 4; ++q) gb[q] = gt_box[q] / norm;
    cross_iou_bbox_row(p, t, act, an, gb, alpha, eps, want_grad, out);
    if (want_grad)
        for (int c = 0; c < 20; ++c) out->grad[c] = out->grad[c] * stride / norm;
}

// ---- polygon (instance segmentation: nv contour vectors + centre) and keypoint (pose: nv keypoints + centre) rows ------
// cross_iou_loss.py:68-77 (polygon: the landmarks are dealt into `sub` interleaved subsets k % sub, the overlap is the
// mean of the subsets' sum(min) / sum(max); box = bounding box of the nv vectors, same distance / aspect terms as the
// bbox row) and :80-94 (keypoint: per (neg, pos) pair the ratio sum(min) / sum(max clamped at eps), weighted by the
// keypoint's visibility, the centre pairs always counted; no box terms).  M = 4 (nv + 1) components per row; nothing
// is copied into local arrays (M = 148 / 72): the row is read where it lies and the gradient written where it goes.
#define LSN_CIOU_MAXSUB 16

// completed target of component c (the inactive half of a pair is alpha x the active half)
LSN_HD float ciou_tt(const float *t, const unsigned char *active, int c, float alpha)
{
    const int j = c & ~1;
    const float act = active[j] ? t[j] : t[j + 1];
    return active[c] ? t[c] : alpha * act;
}

// d min(p, t) / dp and d max(p, t) / dp with ATen's 0.5 / 0.5 split at a tie
LSN_HD float ciou_dmin(float p, float t) { return p < t ? 1.f : (p == t ? .5f : 0.f); }
LSN_HD float ciou_dmax(float p, float t) { return p > t ? 1.f : (p == t ? .5f : 0.f); }

// returns the unweighted loss; grad (M floats, may be NULL) receives d loss / d pred
LSN_HD float cross_iou_polygon_row(const float *p, const float *t, const unsigned char *active, const float *anchor,
                                   const float *gt, int nv, int sub, float alpha, float eps, float *grad)
{
    const int M = 4 * (nv + 1);
    float smin[LSN_CIOU_MAXSUB], smax[LSN_CIOU_MAXSUB];
    for (int s = 0; s < sub; ++s) smin[s] = smax[s] = 0.f;
    for (int k = 0; k <= nv; ++k) {
        const int s = k % sub;
        for (int q = 0; q < 4; ++q) {
            const int c = 4 * k + q;
            const float tt = ciou_tt(t, active, c, alpha);
            smin[s] += fminf(p[c], tt);
            smax[s] += fmaxf(p[c], tt);
        }
    }
    float ov = 0.f;
    for (int s = 0; s < sub; ++s) ov += smin[s] / smax[s];
    ov /= (float)sub;
    // bounding box of the nv vectors (the centre landmark is excluded); first extreme wins a tie
    int kx0 = 0, kx1 = 0, ky0 = 0, ky1 = 0;
    float bx0 = 0.f, bx1 = 0.f, by0 = 0.f, by1 = 0.f;
    for (int k = 0; k < nv; ++k) {
        const float y = ciou_signed(p[4 * k], p[4 * k + 1]) + anchor[1], x = ciou_signed(p[4 * k + 2], p[4 * k + 3]) + anchor[0];
        if (k == 0 || x < bx0) bx0 = x, kx0 = k;
        if (k == 0 || x > bx1) bx1 = x, kx1 = k;
        if (k == 0 || y < by0) by0 = y, ky0 = k;
        if (k == 0 || y > by1) by1 = y, ky1 = k;
    }
    const float ew_raw = fmaxf(bx1, gt[2]) - fminf(bx0, gt[0]), eh_raw = fmaxf(by1, gt[3]) - fminf(by0, gt[1]);
    const float ew = fmaxf(ew_raw, 0.f), eh = fmaxf(eh_raw, 0.f);
    const float c2 = ew * ew + eh * eh + eps;
    const float w1 = bx1 - bx0, h1 = by1 - by0 + eps;
    const float w2 = gt[2] - gt[0], h2 = gt[3] - gt[1] + eps;
    const float dxs = (gt[0] + gt[2]) - (bx0 + bx1), dys = (gt[1] + gt[3]) - (by0 + by1);
    const float rho2 = dxs * dxs / 4 + dys * dys / 4;
    const float kv = 4.f / (3.14159265358979323846f * 3.14159265358979323846f);
    const float da = atanf(w2 / h2) - atanf(w1 / h1);
    const float v = kv * da * da;
    const float D = 1.f - ov + v;
    const float loss = 1.f - (ov - (rho2 / c2 + v * v / D));
    if (!grad) return loss;

    const float dL_dov = -1.f + v * v / (D * D);
    const float dL_dv = (2.f * v * D - v * v) / (D * D);
    const float dL_drho2 = 1.f / c2, dL_dc2 = -rho2 / (c2 * c2);
    const float r = w1 / h1, dA1 = 1.f / (1.f + r * r);
    const float dv_dw1 = kv * 2.f * da * (-dA1 / h1), dv_dh1 = kv * 2.f * da * (dA1 * w1 / (h1 * h1));
    const float cw = ew_raw >= 0.f ? 1.f : 0.f, ch = eh_raw >= 0.f ? 1.f : 0.f;
    const float mx1 = bx1 > gt[2] ? 1.f : (bx1 == gt[2] ? .5f : 0.f), mn0 = bx0 < gt[0] ? 1.f : (bx0 == gt[0] ? .5f : 0.f);
    const float my1 = by1 > gt[3] ? 1.f : (by1 == gt[3] ? .5f : 0.f), my0 = by0 < gt[1] ? 1.f : (by0 == gt[1] ? .5f : 0.f);
    const float g_bx0 = dL_drho2 * (-dxs / 2) + dL_dc2 * 2.f * ew * cw * (-mn0) + dL_dv * dv_dw1 * (-1.f);
    const float g_bx1 = dL_drho2 * (-dxs / 2) + dL_dc2 * 2.f * ew * cw * mx1 + dL_dv * dv_dw1;
    const float g_by0 = dL_drho2 * (-dys / 2) + dL_dc2 * 2.f * eh * ch * (-my0) + dL_dv * dv_dh1 * (-1.f);
    const float g_by1 = dL_drho2 * (-dys / 2) + dL_dc2 * 2.f * eh * ch * my1 + dL_dv * dv_dh1;
    for (int c = 0; c < M; ++c) {
        const int s = (c >> 2) % sub;
        const float tt = ciou_tt(t, active, c, alpha);
        grad[c] = dL_dov / (float)sub * (ciou_dmin(p[c], tt) / smax[s] - smin[s] / (smax[s] * smax[s]) * ciou_dmax(p[c], tt));
    }
    // box coordinates -> the selected half of the pair of their extreme landmark
    const int pair_of[4] = {4 * kx0 + 2, 4 * ky0, 4 * kx1 + 2, 4 * ky1};
    const float g_box[4] = {g_bx0, g_by0, g_bx1, g_by1};
    for (int q = 0; q < 4; ++q) {
        const int c = pair_of[q];
        if (p[c + 1] > p[c]) grad[c + 1] += g_box[q]; else grad[c] -= g_box[q];
    }
    return loss;
}

// vs: nv visibility values (> 0 = counted)
LSN_HD float cross_iou_keypoint_row(const float *p, const float *t, const unsigned char *active, const float *vs, int nv,
                                    float alpha, float eps, float *grad)
{
    const int NP = 2 * (nv + 1);   // (neg, pos) pairs
    float sum = 0.f;
    for (int j = 0; j < NP; ++j) {
        const int c = 2 * j, k = j >> 1;
        const float vis = (k < nv) ? (vs[k] > 0.f ? 1.f : 0.f) : 1.f;
        const float t0 = ciou_tt(t, active, c, alpha), t1 = ciou_tt(t, active, c + 1, alpha);
        const float m0 = fmaxf(p[c], t0), m1 = fmaxf(p[c + 1], t1);
        const float smn = fminf(p[c], t0) + fminf(p[c + 1], t1);
        const float smx = fmaxf(m0, eps) + fmaxf(m1, eps);
        sum += vis * smn / smx;
        if (grad) {
            // clamp(min = eps) passes the gradient where the value is >= eps
            const float d0 = (m0 >= eps ? 1.f : 0.f) * ciou_dmax(p[c], t0), d1 = (m1 >= eps ? 1.f : 0.f) * ciou_dmax(p[c + 1], t1);
            const float sc = -vis / (float)NP;
            grad[c] = sc * (ciou_dmin(p[c], t0) / smx - smn / (smx * smx) * d0);
            grad[c + 1] = sc * (ciou_dmin(p[c + 1], t1) / smx - smn / (smx * smx) * d1);
        }
    }
    return 1.f - sum / (float)NP;
}
