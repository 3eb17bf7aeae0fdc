// Fused cross-IOU loss of the bbox task: one thread per point evaluates the whole row function (target completion,
// overlap, box decode, distance and aspect terms) and, in the backward kernel, its hand-derived gradient
// (cross_iou_row.h; checked against autograd on the CPU, tests/test_fused_cross_iou.py).  Replaces ~60 elementwise
// ATen launches per loss evaluation (and about twice that in backward) by one launch each way; 208 B of reads and
// 4 B (forward) / 80 B (backward) of writes per point: HBM-bound, 45 k points per stage.
#include "common.h"
#include "cross_iou_row.h"

namespace lsn {

struct CiouArgs {
    const float *pred, *target, *anchor, *gt, *weight, *grad_rows;
    const unsigned char *active;
    float *loss, *grad_pred;
    long long n;
    float alpha, eps;
};

template <bool BWD>
__global__ void cross_iou_bbox_kernel(CiouArgs a)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < a.n; i += (long long)gridDim.x * blockDim.x) {
        float p[20], t[20];
        unsigned char act[20];
        const float4 *p4 = reinterpret_cast<const float4 *>(a.pred + 20 * i);
        const float4 *t4 = reinterpret_cast<const float4 *>(a.target + 20 * i);
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const float4 x = p4[q], y = t4[q];
            p[4 * q] = x.x; p[4 * q + 1] = x.y; p[4 * q + 2] = x.z; p[4 * q + 3] = x.w;
            t[4 * q] = y.x; t[4 * q + 1] = y.y; t[4 * q + 2] = y.z; t[4 * q + 3] = y.w;
        }
#pragma unroll
        for (int c = 0; c < 20; ++c) act[c] = a.active[20 * i + c];
        CrossIouRow r;
        cross_iou_bbox_row(p, t, act, a.anchor + 2 * i, a.gt + 4 * i, a.alpha, a.eps, BWD ? 1 : 0, &r);
        const float w = a.weight ? a.weight[i] : 1.f;
        if (!BWD) {
            a.loss[i] = r.loss * w;
        } else {
            const float g = a.grad_rows[i] * w;
            float4 *o = reinterpret_cast<float4 *>(a.grad_pred + 20 * i);
#pragma unroll
            for (int q = 0; q < 5; ++q)
                o[q] = make_float4(r.grad[4 * q] * g, r.grad[4 * q + 1] * g, r.grad[4 * q + 2] * g, r.grad[4 * q + 3] * g);
        }
    }
}

// the whole regression stage per point: raw prediction + extreme points + anchor (x, y, stride) + box -> weighted row
struct CiouStageArgs {
    const float *raw, *gt_pts, *anchor3, *gt_box, *weight, *grad_rows;
    float *loss, *grad_raw;
    long long n;
    float base, alpha, eps;
};

template <bool BWD>
__global__ void cross_iou_bbox_stage_kernel(CiouStageArgs a)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < a.n; i += (long long)gridDim.x * blockDim.x) {
        float p[20], g[10];
        const float4 *p4 = reinterpret_cast<const float4 *>(a.raw + 20 * i);
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const float4 x = p4[q];
            p[4 * q] = x.x; p[4 * q + 1] = x.y; p[4 * q + 2] = x.z; p[4 * q + 3] = x.w;
        }
#pragma unroll
        for (int c = 0; c < 10; ++c) g[c] = a.gt_pts[10 * i + c];
        const float w = a.weight[i];
        CrossIouRow r;
        cross_iou_bbox_stage_row(p, g, a.anchor3 + 3 * i, a.gt_box + 4 * i, w > 0.f, a.base, a.alpha, a.eps, BWD ? 1 : 0, &r);
        if (!BWD) {
            a.loss[i] = r.loss * w;
        } else {
            const float s = a.grad_rows[i] * w;
            float4 *o = reinterpret_cast<float4 *>(a.grad_raw + 20 * i);
#pragma unroll
            for (int q = 0; q < 5; ++q)
                o[q] = make_float4(r.grad[4 * q] * s, r.grad[4 * q + 1] * s, r.grad[4 * q + 2] * s, r.grad[4 * q + 3] * s);
        }
    }
}

// polygon / keypoint rows (cross_iou_row.h): one thread per point, the row read in place (M = 4 (nv + 1) floats)
struct CiouGenArgs {
    const float *pred, *target, *anchor, *gt, *vs, *weight, *grad_rows;
    const unsigned char *active;
    float *loss, *grad_pred;
    long long n;
    int kind, nv, sub;   // kind 1: polygon, 2: keypoint
    float alpha, eps;
};

template <bool BWD>
__global__ void cross_iou_rows_kernel(CiouGenArgs a)
{
    const int M = 4 * (a.nv + 1);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < a.n; i += (long long)gridDim.x * blockDim.x) {
        const float *p = a.pred + M * i, *t = a.target + M * i;
        const unsigned char *act = a.active + M * i;
        float *g = BWD ? a.grad_pred + M * i : nullptr;
        const float w = a.weight ? a.weight[i] : 1.f;
        float loss;
        if (a.kind == 1)
            loss = cross_iou_polygon_row(p, t, act, a.anchor + 2 * i, a.gt + 4 * i, a.nv, a.sub, a.alpha, a.eps, g);
        else
            loss = cross_iou_keypoint_row(p, t, act, a.vs + (long long)a.nv * i, a.nv, a.alpha, a.eps, g);
        if (!BWD) {
            a.loss[i] = loss * w;
        } else {
            const float sc = a.grad_rows[i] * w;
            for (int c = 0; c < M; ++c) g[c] *= sc;
        }
    }
}

static int launch_rows(const CiouGenArgs &a, bool bwd, hipStream_t st)
{
    if (a.n == 0) return 0;
    const int blocks = (int)((a.n + 63) / 64 < 8192 ? (a.n + 63) / 64 : 8192);
    if (bwd) hipLaunchKernelGGL(cross_iou_rows_kernel<true>, dim3(blocks), dim3(64), 0, st, a);
    else hipLaunchKernelGGL(cross_iou_rows_kernel<false>, dim3(blocks), dim3(64), 0, st, a);
    LSN_HIP(hipGetLastError());
    return 0;
}

static int check_rows(const char *who, int kind, int64_t n, int ncomp, int sub, const void *pred, const void *target,
                      const void *active, const void *anchor, const void *gt, const void *vs)
{
    LSN_CHECK(n >= 0, "%s: invalid number of rows %lld", who, (long long)n);
    LSN_CHECK(kind == 1 || kind == 2, "%s: kind must be 1 (polygon) or 2 (keypoint), got %d", who, kind);
    LSN_CHECK(ncomp >= 8 && ncomp % 4 == 0, "%s: rows have 4 (nv + 1) components, got %d", who, ncomp);
    if (n == 0) return 0;
    LSN_CHECK(pred && target && active, "%s: null pointer", who);
    if (kind == 1) {
        LSN_CHECK(anchor && gt, "%s: the polygon loss needs anchor points and ground-truth boxes", who);
        LSN_CHECK(sub >= 1 && sub <= LSN_CIOU_MAXSUB && sub <= ncomp / 4, "%s: subset stride %d out of range", who, sub);
    } else {
        LSN_CHECK(vs != nullptr, "%s: the keypoint loss needs the visibility values", who);
    }
    return 0;
}

static int launch(const CiouArgs &a, bool bwd, hipStream_t st)
{
    if (a.n == 0) return 0;
    const int blocks = (int)((a.n + 255) / 256 < 4096 ? (a.n + 255) / 256 : 4096);
    if (bwd) hipLaunchKernelGGL(cross_iou_bbox_kernel<true>, dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(cross_iou_bbox_kernel<false>, dim3(blocks), dim3(256), 0, st, a);
    LSN_HIP(hipGetLastError());
    return 0;
}

}  // namespace lsn

extern "C" {

int lsn_cross_iou_bbox_forward(const float *pred, const float *target, const uint8_t *active, const float *anchor,
                               const float *bbox_gt, const float *weight, int64_t n, float alpha, float eps,
                               float *loss_rows, lsn_stream_t stream)
{
    using namespace lsn;
    LSN_CHECK(n >= 0, "invalid number of rows %lld", (long long)n);
    LSN_CHECK(n == 0 || (pred && target && active && anchor && bbox_gt && loss_rows), "lsn_cross_iou_bbox_forward: null pointer");
    LSN_CHECK(((uintptr_t)pred % 16 == 0) && ((uintptr_t)target % 16 == 0), "pred / target rows must be 16-byte aligned");
    CiouArgs a = {pred, target, anchor, bbox_gt, weight, nullptr, active, loss_rows, nullptr, (long long)n, alpha, eps};
    return launch(a, false, static_cast<hipStream_t>(stream));
}

int lsn_cross_iou_bbox_backward(const float *pred, const float *target, const uint8_t *active, const float *anchor,
                                const float *bbox_gt, const float *weight, const float *grad_rows, int64_t n,
                                float alpha, float eps, float *grad_pred, lsn_stream_t stream)
{
    using namespace lsn;
    LSN_CHECK(n >= 0, "invalid number of rows %lld", (long long)n);
    LSN_CHECK(n == 0 || (pred && target && active && anchor && bbox_gt && grad_rows && grad_pred),
              "lsn_cross_iou_bbox_backward: null pointer");
    LSN_CHECK(((uintptr_t)pred % 16 == 0) && ((uintptr_t)target % 16 == 0) && ((uintptr_t)grad_pred % 16 == 0),
              "pred / target / grad rows must be 16-byte aligned");
    CiouArgs a = {pred, target, anchor, bbox_gt, weight, grad_rows, active, nullptr, grad_pred, (long long)n, alpha, eps};
    return launch(a, true, static_cast<hipStream_t>(stream));
}

int lsn_cross_iou_bbox_stage_forward(const float *pred_raw, const float *gt_pts, const float *anchor3, const float *bbox_gt,
                                     const float *weight, int64_t n, float base_scale, float alpha, float eps,
                                     float *loss_rows, lsn_stream_t stream)
{
    using namespace lsn;
    LSN_CHECK(n >= 0, "invalid number of rows %lld", (long long)n);
    if (n == 0) return 0;
    LSN_CHECK(pred_raw && gt_pts && anchor3 && bbox_gt && weight && loss_rows, "lsn_cross_iou_bbox_stage_forward: null pointer");
    LSN_CHECK((uintptr_t)pred_raw % 16 == 0, "prediction rows must be 16-byte aligned");
    CiouStageArgs a = {pred_raw, gt_pts, anchor3, bbox_gt, weight, nullptr, loss_rows, nullptr, (long long)n, base_scale, alpha, eps};
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(cross_iou_bbox_stage_kernel<false>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    LSN_HIP(hipGetLastError());
    return 0;
}

int lsn_cross_iou_bbox_stage_backward(const float *pred_raw, const float *gt_pts, const float *anchor3, const float *bbox_gt,
                                      const float *weight, const float *grad_rows, int64_t n, float base_scale, float alpha,
                                      float eps, float *grad_raw, lsn_stream_t stream)
{
    using namespace lsn;
    LSN_CHECK(n >= 0, "invalid number of rows %lld", (long long)n);
    if (n == 0) return 0;
    LSN_CHECK(pred_raw && gt_pts && anchor3 && bbox_gt && weight && grad_rows && grad_raw,
              "lsn_cross_iou_bbox_stage_backward: null pointer");
    LSN_CHECK((uintptr_t)pred_raw % 16 == 0 && (uintptr_t)grad_raw % 16 == 0, "prediction / gradient rows must be 16-byte aligned");
    CiouStageArgs a = {pred_raw, gt_pts, anchor3, bbox_gt, weight, grad_rows, nullptr, grad_raw, (long long)n, base_scale, alpha, eps};
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(cross_iou_bbox_stage_kernel<true>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    LSN_HIP(hipGetLastError());
    return 0;
}

int lsn_cross_iou_rows_forward(int kind, const float *pred, const float *target, const uint8_t *active, const float *anchor,
                               const float *bbox_gt, const float *vs, const float *weight, int64_t n, int ncomp, int sub,
                               float alpha, float eps, float *loss_rows, lsn_stream_t stream)
{
    using namespace lsn;
    if (int rc = check_rows("lsn_cross_iou_rows_forward", kind, n, ncomp, sub, pred, target, active, anchor, bbox_gt, vs)) return rc;
    LSN_CHECK(n == 0 || loss_rows, "lsn_cross_iou_rows_forward: null output");
    CiouGenArgs a = {pred, target, anchor, bbox_gt, vs, weight, nullptr, active, loss_rows, nullptr, (long long)n, kind,
                     ncomp / 4 - 1, sub, alpha, eps};
    return launch_rows(a, false, static_cast<hipStream_t>(stream));
}

int lsn_cross_iou_rows_backward(int kind, const float *pred, const float *target, const uint8_t *active, const float *anchor,
                                const float *bbox_gt, const float *vs, const float *weight, const float *grad_rows, int64_t n,
                                int ncomp, int sub, float alpha, float eps, float *grad_pred, lsn_stream_t stream)
{
    using namespace lsn;
    if (int rc = check_rows("lsn_cross_iou_rows_backward", kind, n, ncomp, sub, pred, target, active, anchor, bbox_gt, vs)) return rc;
    LSN_CHECK(n == 0 || (grad_rows && grad_pred), "lsn_cross_iou_rows_backward: null pointer");
    CiouGenArgs a = {pred, target, anchor, bbox_gt, vs, weight, grad_rows, active, nullptr, grad_pred, (long long)n, kind,
                     ncomp / 4 - 1, sub, alpha, eps};
    return launch_rows(a, true, static_cast<hipStream_t>(stream));
}

}  // extern "C"
