/* lsnet_host.h -- C ABI of liblsnet_host.so: host-side (CPU) helpers of the evaluation path.
 *
 * Replaces, for a maintainer of the reference, the compiled half of its vendored COCO api: the run-length mask
 * routines of cocoapi/pycocotools/common/maskApi.c that `pycocotools._mask` binds (cocoapi/pycocotools/pycocotools/
 * _mask.pyx:96-308) and that evaluation reaches through mmdet/core/mask/utils.py:65-68 (polygon -> RLE of the segm
 * task's results) and pycocotools/cocoeval.py:126-152 (IoU matrices).  Plain pointers and sizes; no Python, no torch.
 *
 * A run-length mask ("RLE") of an h x w grid is an array of uint32 run lengths over the column-major pixel order
 * (index = x * h + y), alternating background / foreground and starting with background (a leading 0 when pixel 0
 * is foreground).  Lists of masks are passed flattened: `counts` holds all runs back to back, mask i owns
 * counts[offsets[i] .. offsets[i+1]).
 *
 * Functions that produce a mask return the number of runs m; they write at most `cap` runs.  m > cap means the
 * buffer was too small: call again with a larger one.  Results are identical, run for run, to the reference's.
 */
#ifndef LSNET_HOST_H
#define LSNET_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* maskApi.c:162-202 rleFrPoly: scan-convert a closed polygon xy = [x0,y0,x1,y1,...] (k vertices, pixel coordinates,
 * COCO's rasterisation rule: 5x super-sampled boundary walk, column crossings, downsample). */
size_t lsn_rle_from_polygon(const double *xy, size_t k, uint32_t h, uint32_t w, uint32_t *counts, size_t cap);

/* maskApi.c:149-156 rleFrBbox: the polygon of the box [x, y, w, h]. */
size_t lsn_rle_from_bbox(const double *bbox, uint32_t h, uint32_t w, uint32_t *counts, size_t cap);

/* maskApi.c:49-70 rleMerge: union (intersect = 0) or intersection (1) of n masks of one grid. */
size_t lsn_rle_merge(const uint32_t *counts, const size_t *offsets, size_t n, int intersect, uint32_t *out,
                     size_t cap);

/* maskApi.c:72-75 rleArea: foreground pixels of each mask. */
void lsn_rle_area(const uint32_t *counts, const size_t *offsets, size_t n, uint32_t *area);

/* maskApi.c:133-147 rleToBbox: [x, y, w, h] of each mask (zeros for an empty one). */
void lsn_rle_to_bbox(const uint32_t *counts, const size_t *offsets, size_t n, const uint32_t *hs, const uint32_t *ws,
                     double *bbox);

/* maskApi.c:77-96 rleIou: out[d * n + g] = |dt_d & gt_g| / |dt_d | gt_g|, or / |dt_d| where iscrowd[g]; -1 for
 * grids of different size whose boxes overlap.  `iscrowd` may be NULL. */
void lsn_rle_iou(const uint32_t *dt_counts, const size_t *dt_offsets, const uint32_t *dt_h, const uint32_t *dt_w,
                 size_t m, const uint32_t *gt_counts, const size_t *gt_offsets, const uint32_t *gt_h,
                 const uint32_t *gt_w, size_t n, const uint8_t *iscrowd, double *out);

/* maskApi.c:109-120 bbIou: boxes [x, y, w, h]; same output layout and crowd rule. */
void lsn_bbox_iou(const double *dt, size_t m, const double *gt, size_t n, const uint8_t *iscrowd, double *out);

/* maskApi.c:32-47 rleEncode / rleDecode of ONE column-major (Fortran-order) h x w byte mask. */
size_t lsn_rle_encode(const uint8_t *mask, uint32_t h, uint32_t w, uint32_t *counts, size_t cap);
void lsn_rle_decode(const uint32_t *counts, size_t m, uint8_t *mask, size_t hw);

/* maskApi.c:204-231 rleToString / rleFrString: COCO's compressed ASCII form.  to_string returns the string length
 * (without the terminating 0, which is written when it fits); from_string returns the number of runs. */
size_t lsn_rle_to_string(const uint32_t *counts, size_t m, char *s, size_t cap);
size_t lsn_rle_from_string(const char *s, uint32_t *counts, size_t cap);

/* cocoeval.py:212-247 (COCOeval.evaluateImg, the matching loops): greedy assignment of detections (rows of `ious`,
 * D x G row-major, sorted by descending score) to ground truths (columns, non-ignored first) for T IoU thresholds.
 * dt_match[t*D + d] = matched column or -1; gt_match[t*G + g] = matching row or -1. */
void lsn_coco_match(const double *ious, size_t D, size_t G, const uint8_t *gt_ignore, const uint8_t *gt_crowd,
                    const double *thrs, size_t T, int64_t *dt_match, int64_t *gt_match);

/* Image preparation of the data pipeline; the reference reaches OpenCV for these through mmcv
 * (mmcv/image/geometric.py:26-56 imresize, photometric.py:8-41 imnormalize).  Images are interleaved h x w x c.
 * Bilinear resize with cv2.INTER_LINEAR's sampling (pixel centres at half-integers, no antialiasing); 8-bit images
 * use its 11-bit fixed-point weights and two-pass rounding, float images plain float32.  Return 0, or 1 on a
 * non-positive size. */
int lsn_image_resize_bilinear_u8(const uint8_t *src, int sh, int sw, int c, uint8_t *dst, int dh, int dw);
int lsn_image_resize_bilinear_f32(const float *src, int sh, int sw, int c, float *dst, int dh, int dw);

/* dst = (src - mean) * inv_std per channel in float32, optionally reading the channels in reverse order (BGR -> RGB). */
void lsn_image_normalize_u8(const uint8_t *src, size_t pixels, int c, const float *mean, const float *inv_std,
                            int reverse_channels, float *dst);
void lsn_image_normalize_f32(const float *src, size_t pixels, int c, const float *mean, const float *inv_std,
                             int reverse_channels, float *dst);

/* The two host-only members of the reference's `nms_ext` module (mmdet/ops/nms/src/nms_ext.cpp:29-43, cpu/nms_cpu.cpp:
 * 63-258); `dets` is (n, 5) float32 [x1, y1, x2, y2, score].
 * lsn_soft_nms: method 1 linear, 2 gaussian, other = hard; out has room for n rows [x1, y1, x2, y2, score, index];
 *               returns the number of rows written.
 * lsn_nms_match: `order` = box indices by descending score; flat (n) receives the groups back to back (keeper first),
 *               group_start (n + 1) their boundaries; returns the number of groups. */
size_t lsn_soft_nms(const float *dets, size_t n, float iou_thr, int method, float sigma, float min_score, float *out);
/* Hard NMS of host tensors: `nms_cpu` of the reference's dispatcher (nms_ext.cpp:11-27 -> cpu/nms_cpu.cpp:7-71), for
 * float32 and float64 boxes as AT_DISPATCH_FLOATING_TYPES does.  `order` = box indices by descending score (the caller
 * sorts, as the reference does with torch.sort); keep (n) receives the kept indices in visiting order; returns their
 * number.  A box goes when its IoU with a kept box EXCEEDS iou_thr. */
size_t lsn_nms_host_f32(const float *dets, const int64_t *order, size_t n, float iou_thr, int64_t *keep);
size_t lsn_nms_host_f64(const double *dets, const int64_t *order, size_t n, float iou_thr, int64_t *keep);
size_t lsn_nms_match(const float *dets, const int64_t *order, size_t n, float iou_thr, int64_t *flat, int64_t *group_start);

#ifdef __cplusplus
}
#endif
#endif
