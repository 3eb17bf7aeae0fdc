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

}  // extern "C"
