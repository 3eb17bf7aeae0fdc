/*
 * lsnet_hip.h -- C ABI of liblsnet_hip.so: the MI355X (gfx950) replacement for the reference's
 * native-op boundary on the LSNet hot path.
 *
 * What it replaces (all paths relative to /root/reference/code):
 *   mmdet/ops/dcn/src/deform_conv_ext.cpp:227-250      pybind module `deform_conv_ext` (8 functions)
 *   mmdet/ops/sigmoid_focal_loss/src/sigmoid_focal_loss_ext.cpp:19-57   `sigmoid_focal_loss_ext`
 *   mmdet/ops/nms/src/nms_ext.cpp:18-49                 `nms_ext.nms`
 * plus device kernels for work the reference gets from PyTorch (dense convolution, GroupNorm + ReLU, frozen
 * BatchNorm + add + ReLU, weighted focal-loss sums) -- marked [fused] or described in their sections below.
 *
 * Conventions
 *   - plain C: raw device pointers, int sizes, an explicit hipStream_t.  No torch types.
 *   - every function returns 0 on success, a negative lsn_status otherwise; lsn_last_error()
 *     returns a thread-local message (the reference raises RuntimeError from TORCH_CHECK,
 *     deform_conv_cuda.cpp:92-180,182-272; the Python mirror raises RuntimeError with this text).
 *   - the caller owns all buffers (as in the reference, deform_conv.py:40-42,141-142,215-217).
 *     Outputs and gradient buffers are OVERWRITTEN (the reference's Python always passes
 *     zero-filled gradient tensors, deform_conv.py:75-76,86,157-161, so results are identical).
 *   - work is enqueued on `stream` and is asynchronous w.r.t. the host, except lsn_nms.
 *   - float32 tensors throughout; indices int64 where the reference uses int64.  The arithmetic of the
 *     contractions is selected by lsn_set_math_mode (split-bf16 products with fp32 accumulation, or exact fp32).
 *   - one process per GPU; entry points are re-entrant across streams.  Process-wide mutable state: the math
 *     mode, the kernel-timing log (lsn_prof_*) and the tuning word (lsn_debug_phase_clocks) -- set them from one
 *     thread; plus the thread-local error string.
 *
 * Activation layout.  `layout` selects how 4-D activation tensors (input, output, their grads)
 * and the weight tensor are laid out in memory:
 *     LSN_NCHW : input (B,C,H,W) contiguous, weight (Co,C/g,kh,kw) contiguous -- exactly what the
 *                reference extension receives.  Handled by device-side permutes around the NHWC
 *                kernels (a stream-ordered workspace is allocated with hipMallocAsync).
 *     LSN_NHWC : the same logical tensors in channels-last memory (B,H,W,C) / (Co,kh,kw,C/g).  This
 *                is the native layout of the kernels: the bilinear gather reads contiguous channel
 *                vectors, so every global access of a wavefront is a coalesced 128-256 B segment.
 * Offset / mask tensors (and their grads) are described by explicit element strides
 * (lsn_strides4), so either memory format works without a copy.
 */
#ifndef LSNET_HIP_H_
#define LSNET_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t *lsn_stream_t; /* == hipStream_t */

typedef enum {
    LSN_OK = 0,
    LSN_ERR_INVALID = -1,     /* shape / argument check failed (TORCH_CHECK in the reference) */
    LSN_ERR_UNSUPPORTED = -2, /* valid in the reference, not implemented by this build        */
    LSN_ERR_RUNTIME = -3      /* HIP runtime error (launch, allocation)                        */
} lsn_status;

typedef enum { LSN_NCHW = 0, LSN_NHWC = 1 } lsn_layout;

/* element strides of a logical (B, Ch, H, W) tensor */
typedef struct { int64_t b, c, h, w; } lsn_strides4;

/* One deformable-convolution call site.  (H,W) is the sampled source map; (Ho,Wo) the output
 * grid, which for the pyramid op is the OFFSET grid (deform_conv.py:215-217).
 * stride/pad/dil are the same for h and w (deform_conv.py:146-148).
 * scale_h/scale_w: 1 for DCNv1/v2; the pyramid op samples at
 *      (ho*stride - pad + i*dil) * scale_h + dy      (deform_conv_cuda_kernel.cu:281-282). */
typedef struct {
    int B, C, H, W;
    int Co, Ho, Wo;
    int kh, kw, stride, pad, dil;
    int groups, deformable_groups;
    float scale_h, scale_w;
    int mask_is_logit; /* 1: `mask` holds raw logits; the kernels apply sigmoid and grad_mask is the
                        * gradient w.r.t. the logits.  Lets a DCNv2 pack hand its (B, 3*dg*kh*kw, H, W)
                        * conv_offset output to the op as ONE tensor (offset = first 2/3 of the channels,
                        * mask = last 1/3, deform_conv.py:527-530) and get ONE gradient tensor back. */
    void *workspace;   /* optional device scratch of >= Co*kh*kw*(C/groups)*8 bytes: in the split-bf16 math modes
                        * the call's weight image goes there (MFMA fragment order for the shapes the kernels of
                        * csrc/dcn_mm_kernels.h serve, pre-split -- backward: also transposed -- bf16 planes
                        * otherwise) and the matrix-pipe kernels run; NULL: weights are split inside every block
                        * (forward) / the fp32 MFMA backward-data kernel runs.  Contents undefined afterwards. */
    void *gather_workspace;          /* optional device scratch for lsn_dcn_backward's atomic-free grad_input path */
    int64_t gather_workspace_bytes;  /* (>= lsn_dcn_backward_workspace_bytes()); NULL / too small: fp32 atomics.   */
    int accumulate_param_grads;      /* lsn_dcn_backward: 1 = ADD grad_weight / grad_bias to the buffers' contents (see
                                      * lsn_conv2d_backward_weight), 0 = overwrite */
    int out_pitch;                   /* LSN_NHWC only: floats from one pixel of `output` (forward) / `grad_output`
                                      * (backward) to the next, for every level; 0 = Co (dense).  With Co < out_pitch the
                                      * levels' outputs are channel slices of wider tensors: LSHead concatenates the three
                                      * maps a level gathers from its neighbours (lsnet_head.py:640-647) -- the op writes them
                                      * side by side and reads their gradient where the next convolution left it, no
                                      * concatenation or slice copy in between.  Served by the matrix-pipe kernels only:
                                      * ask lsn_dcn_pitched_ok() first, the calls fail with LSN_ERR_UNSUPPORTED otherwise. */
    int weights_prepared;            /* 1: `workspace` already HOLDS the weight's fragment image for this pass -- forward: the
                                      * image lsn_conv2d_prepare_weights(kind 0) builds of the (Co, kh, kw, C) weight; backward:
                                      * kind 2 -- and the call does not rebuild it (round 6: the head's deformable layers keep
                                      * their images per optimizer step like the dense convolutions, 16 launches less per step;
                                      * the reference re-reads its weight in every call, deform_conv_cuda.cpp:662-684).  Only for
                                      * calls the matrix-pipe kernels serve: ask lsn_dcn_prepared_ok() first, the call fails
                                      * with LSN_ERR_UNSUPPORTED otherwise.  The image is read, not modified. */
} lsn_dcn_shape;

/* One (source map, offset field, output) triple of a batched launch.  All levels of a launch
 * share weight/bias and the conv hyper-parameters; B,H,W,Ho,Wo,scale_* may differ per level.
 * This is how the five FPN levels of LSHead's shared towers (lsnet_head.py:502-513) and the
 * 15 (dst,src) pairs of one PyramidDeformConv (lsnet_head.py:622-638) become ONE launch. */
typedef struct {
    const float *input;      /* (B,C,H,W) in `layout`                                  */
    const float *offset;     /* (B, dg*2*kh*kw, Ho, Wo), strides off_st                */
    const float *mask;       /* (B, dg*kh*kw, Ho, Wo), strides mask_st; NULL = no mask */
    float *output;           /* (B,Co,Ho,Wo) in `layout`  [forward]                    */
    const float *grad_output;/* (B,Co,Ho,Wo) in `layout`  [backward]                   */
    float *grad_input;       /* may be NULL                                            */
    float *grad_offset;      /* same strides as offset; may be NULL                    */
    float *grad_mask;        /* same strides as mask;   may be NULL                    */
    lsn_strides4 off_st, mask_st;
    int B, H, W, Ho, Wo;
    float scale_h, scale_w;
} lsn_dcn_level;

const char *lsn_last_error(void);
int lsn_version(void);

/* ---- arithmetic of the contractions ---------------------------------------------------------
 * LSN_MATH_FP32   : v_mfma_f32_*_f32, exact fp32 products (bitwise an fmaf chain); fp32 vector rate.
 * LSN_MATH_BF16X6 : (default) fp32-EQUIVALENT.  Every fp32 operand is split exactly into three bf16 values
 *                   (h + m + l, 24 mantissa bits) and a product is h*h + h*m + m*h + m*m + h*l + l*h on the bf16
 *                   matrix pipe with fp32 accumulation; the dropped terms are <= 2^-25 relative, below the 2^-24
 *                   rounding of an fp32 product, so results differ from LSN_MATH_FP32 by summation order only
 *                   (tests: <= 1e-6 of the output range).  2516 / 6 = 419 TFLOP/s peak against 157 for fp32 MFMA.
 * LSN_MATH_BF16X3 : two bf16 values per operand (16 mantissa bits), products h*h + h*l + l*h: relative error
 *                   <= 2^-16 per product (the reference's tolerance for this path is 1e-3).  Opt-in.
 * Process-wide; the initial value comes from the environment variable LSNET_MATH (fp32 | bf16x3 | bf16x6).
 * Kernels without a split variant keep using fp32 MFMA. */
enum { LSN_MATH_FP32 = 0, LSN_MATH_BF16X3 = 1, LSN_MATH_BF16X6 = 2 };
int lsn_set_math_mode(int mode);
int lsn_get_math_mode(void);

/* ---- generic batched entry points (what the Python mirror calls) -------------------------- */

/* out_l = DCN(input_l, offset_l, mask_l; weight, bias) for l < n_levels.
 * shape->B/H/W/Ho/Wo/scale_* are ignored (taken per level); bias may be NULL. */
int lsn_dcn_forward(const lsn_dcn_shape *shape, int n_levels, const lsn_dcn_level *levels,
                    const float *weight, const float *bias, lsn_layout layout, lsn_stream_t stream);

/* Gradients of the above.  grad_weight (layout of weight) and grad_bias (Co) are summed over
 * all levels and images; either may be NULL.  Per level: grad_input / grad_offset / grad_mask. */
int lsn_dcn_backward(const lsn_dcn_shape *shape, int n_levels, const lsn_dcn_level *levels,
                     const float *weight, float *grad_weight, float *grad_bias, lsn_layout layout,
                     lsn_stream_t stream);

/* Bytes of `gather_workspace` that let lsn_dcn_backward form grad_input without atomics (per-anchor sample lists, a
 * column-gradient buffer of sum(B*Ho*Wo) * kh*kw * C floats written and read once, per-anchor corner sums and the
 * corner products of the offset / mask gradients); 0 when that path does not apply to the shape (groups > 1,
 * exact-fp32 mode, no grad_input requested).  Levels whose grad_input pointers are EQUAL accumulate into that one
 * buffer (several offset fields sampling one source map).  With this scratch the whole backward pass -- grad_input,
 * grad_offset, grad_mask, grad_weight, grad_bias -- is free of floating-point atomics in the split-bf16 math modes
 * (groups = 1): bit-identical run to run. */
int64_t lsn_dcn_backward_workspace_bytes(const lsn_dcn_shape *shape, int n_levels, const lsn_dcn_level *levels);
/* 1 when lsn_dcn_forward (backward == 0) / lsn_dcn_backward (backward != 0) would serve this call with
 * shape->out_pitch != Co in the current math mode (shape->workspace / gather_workspace as they will be passed), else 0. */
int lsn_dcn_pitched_ok(const lsn_dcn_shape *shape, int n_levels, const lsn_dcn_level *levels, int backward);
/* 1 when the call would read its weight from a fragment image in shape->workspace (the matrix-pipe kernels: forward, and the
 * backward's column-gradient GEMM), i.e. when shape->weights_prepared may be set; workspace / gather_workspace as they will
 * be passed. */
int lsn_dcn_prepared_ok(const lsn_dcn_shape *shape, int n_levels, const lsn_dcn_level *levels, int backward);

/* ---- one-to-one replacements of the reference extension's functions ----------------------- */
/* Each takes contiguous NCHW tensors like the reference and the same scalar arguments in the
 * same order (note W-before-H for v1/pyramid, H-before-W for the modulated op); the scratch
 * tensors `columns` / `ones` of the reference have no counterpart (nothing is materialised). */

/* deform_conv_ext.cpp:74  deform_conv_forward */
int lsn_deform_conv_forward(const float *input, const float *weight, const float *offset, float *output,
                            int B, int C, int H, int W, int Co, int kW, int kH, int dW, int dH, int padW,
                            int padH, int dilW, int dilH, int group, int deformable_group,
                            int im2col_step, lsn_stream_t stream);
/* deform_conv_ext.cpp:92  deform_conv_backward_input */
int lsn_deform_conv_backward_input(const float *input, const float *offset, const float *grad_output,
                                   float *grad_input, float *grad_offset, const float *weight, int B,
                                   int C, int H, int W, int Co, int kW, int kH, int dW, int dH, int padW,
                                   int padH, int dilW, int dilH, int group, int deformable_group,
                                   int im2col_step, lsn_stream_t stream);
/* deform_conv_ext.cpp:111 deform_conv_backward_parameters (grad_weight = scale * dL/dW) */
int lsn_deform_conv_backward_parameters(const float *input, const float *offset, const float *grad_output,
                                        float *grad_weight, int B, int C, int H, int W, int Co, int kW,
                                        int kH, int dW, int dH, int padW, int padH, int dilW, int dilH,
                                        int group, int deformable_group, float scale, int im2col_step,
                                        lsn_stream_t stream);
/* deform_conv_ext.cpp:129 modulated_deform_conv_forward */
int lsn_modulated_deform_conv_forward(const float *input, const float *weight, const float *bias,
                                      const float *offset, const float *mask, float *output, int B, int C,
                                      int H, int W, int Co, int kernel_h, int kernel_w, int stride_h,
                                      int stride_w, int pad_h, int pad_w, int dilation_h, int dilation_w,
                                      int group, int deformable_group, int with_bias, lsn_stream_t stream);
/* deform_conv_ext.cpp:149 modulated_deform_conv_backward */
int lsn_modulated_deform_conv_backward(const float *input, const float *weight, const float *bias,
                                       const float *offset, const float *mask, float *grad_input,
                                       float *grad_weight, float *grad_bias, float *grad_offset,
                                       float *grad_mask, const float *grad_output, int B, int C, int H,
                                       int W, int Co, int kernel_h, int kernel_w, int stride_h,
                                       int stride_w, int pad_h, int pad_w, int dilation_h, int dilation_w,
                                       int group, int deformable_group, int with_bias,
                                       lsn_stream_t stream);
/* deform_conv_ext.cpp:171 pyramid_deform_conv_forward; (Ho,Wo) = offset grid */
int lsn_pyramid_deform_conv_forward(const float *input, const float *weight, const float *offset,
                                    float *output, int B, int C, int H, int W, int Co, int Ho, int Wo,
                                    int kW, int kH, int dW, int dH, int padW, int padH, int dilW, int dilH,
                                    float scaleW, float scaleH, int group, int deformable_group,
                                    int im2col_step, lsn_stream_t stream);
/* deform_conv_ext.cpp:190 pyramid_deform_conv_backward_input */
int lsn_pyramid_deform_conv_backward_input(const float *input, const float *offset,
                                           const float *grad_output, float *grad_input,
                                           float *grad_offset, const float *weight, int B, int C, int H,
                                           int W, int Co, int Ho, int Wo, int kW, int kH, int dW, int dH,
                                           int padW, int padH, int dilW, int dilH, float scaleW,
                                           float scaleH, int group, int deformable_group,
                                           int im2col_step, lsn_stream_t stream);
/* deform_conv_ext.cpp:209 pyramid_deform_conv_backward_parameters */
int lsn_pyramid_deform_conv_backward_parameters(const float *input, const float *offset,
                                                const float *grad_output, float *grad_weight, int B,
                                                int C, int H, int W, int Co, int Ho, int Wo, int kW,
                                                int kH, int dW, int dH, int padW, int padH, int dilW,
                                                int dilH, float scaleW, float scaleH, int group,
                                                int deformable_group, float scale, int im2col_step,
                                                lsn_stream_t stream);

/* ---- sigmoid focal loss: sigmoid_focal_loss_ext.cpp:19-57 --------------------------------- */
/* logits (N,C) f32 row-major, targets (N) i64 in [0,C] (C = background), losses (N,C). */
int lsn_sigmoid_focal_loss_forward(const float *logits, const int64_t *targets, float *losses, int N,
                                   int C, float gamma, float alpha, lsn_stream_t stream);
int lsn_sigmoid_focal_loss_backward(const float *logits, const int64_t *targets, const float *d_losses,
                                    float *d_logits, int N, int C, float gamma, float alpha,
                                    lsn_stream_t stream);
/* [fused] *loss_sum (device scalar, overwritten) = sum_n weight[n] * sum_c FL(n,c).  Replaces
 * FocalLoss.forward's elementwise-weight + sum chain (focal_loss.py:74-116,
 * losses/utils.py:27-62); the caller divides by avg_factor.  `weight` may be NULL. */
int lsn_sigmoid_focal_loss_sum(const float *logits, const int64_t *targets, const float *weight,
                               float *loss_sum, int N, int C, float gamma, float alpha,
                               lsn_stream_t stream);
/* [fused] d_logits[n,c] = (*scale) * weight[n] * dFL(n,c)/dlogit; `scale` is a DEVICE scalar (the
 * upstream gradient times loss_weight / avg_factor), so no host synchronisation is needed. */
int lsn_sigmoid_focal_loss_backward_weighted(const float *logits, const int64_t *targets,
                                             const float *weight, const float *scale, float *d_logits,
                                             int N, int C, float gamma, float alpha, lsn_stream_t stream);

/* [fused] Per-LEVEL loss sums over LSHead's concatenated rows.  The head keeps the pixels of all FPN levels of an image back to
 * back (row r = b * N_all + i; level l owns i in [level_starts[l], level_starts[l + 1]), level_starts: HOST array of L + 1
 * entries, L <= 8, B * L <= 64).  The reference computes every loss term per level (lsnet_head.py:1021-1270 `loss_single` through
 * `multi_apply`, then base.py:176-209 adds the levels up): these calls return the L per-level values of one term in one launch.
 *   lsn_sigmoid_focal_loss_level_sums: loss_sums[l] = sum over level l's rows of weight[n] * sum_c FL(n, c)  (focal_loss.py:74-116)
 *   lsn_sigmoid_focal_loss_backward_levels: d_logits[n, c] = scales[level(n)] * weight[n] * dFL(n, c)/dlogit  (scales: DEVICE, L)
 *   lsn_level_sums: sums[l] = sum over level l's rows of rows[r]   (the per-row cross-IOU terms)
 *   lsn_level_expand: out_rows[r] = g[level(r)]                    (its gradient)
 * Fixed summation orders: the same bits on every run. */
int lsn_sigmoid_focal_loss_level_sums(const float *logits, const int64_t *targets, const float *weight, float *loss_sums,
                                      int B, int N_all, int C, int L, const int *level_starts, float gamma, float alpha,
                                      lsn_stream_t stream);
int lsn_sigmoid_focal_loss_backward_levels(const float *logits, const int64_t *targets, const float *weight, const float *scales,
                                           float *d_logits, int B, int N_all, int C, int L, const int *level_starts,
                                           float gamma, float alpha, lsn_stream_t stream);
int lsn_level_sums(const float *rows, float *sums, int B, int N_all, int L, const int *level_starts, lsn_stream_t stream);
int lsn_level_expand(const float *g, float *out_rows, int B, int N_all, int L, const int *level_starts, lsn_stream_t stream);

/* ---- gradient clipping + SGD: mmcv/runner/hooks/optimizer.py:8-28 ----------------------------- [fused]
 * clip_grad_norm_(parameters, max_norm, norm_type = 2) followed by torch.optim.SGD.step() (momentum, weight decay; no
 * dampening, no Nesterov) over all parameter tensors in three launches instead of ATen's dozen multi-tensor ones.
 * `tensors_dev`: DEVICE array of n_tensors entries in ascending first_chunk order: a tensor of numel floats owns the
 * ceil(numel / 4096) chunks from first_chunk on (param / grad / momentum_buf: dense tensors of identical strides, 16-byte
 * aligned); total_chunks = their sum.  groups[t.group] = that parameter group's (lr, momentum, weight_decay).
 * max_norm > 0: stats[0] = the total gradient norm, stats[1] = min(1, max_norm / (norm + 1e-6)) and the gradients are
 * scaled by it IN PLACE when it is below 1 (as the reference leaves p.grad); max_norm <= 0: no clipping, stats untouched.
 * Per element every operation is rounded as the operator sequence of torch rounds it (csrc/misc.hip).  A zero-filled
 * momentum buffer makes the first step what torch's is.  `workspace`: lsn_clip_sgd_workspace_bytes() bytes.
 * Everything is device-side: no host read of the norm. */
typedef struct {
    float *param, *grad, *momentum_buf;
    int64_t numel, first_chunk;
    int group;
} lsn_sgd_tensor;
typedef struct {
    float lr, momentum, weight_decay;
} lsn_sgd_group;
int64_t lsn_clip_sgd_workspace_bytes(void);
int lsn_clip_sgd_step(int n_tensors, const lsn_sgd_tensor *tensors_dev, int64_t total_chunks, int n_groups,
                      const lsn_sgd_group *groups, float max_norm, void *workspace, float *stats, lsn_stream_t stream);

/* ---- LSHead's cumulative offset rescaling: lsnet_head.py:622-638 ------------------------------ [fused]
 * The reference multiplies a level's offset field IN PLACE by (scale_h, scale_w) of each of the three source levels it
 * visits, so the pyramid convolutions of a destination level see off*m1, off*m1*m2, off*m1*m2*m3 (y channels by mh, x
 * channels by mw; channel order y0 x0 y1 x1 ...).  One launch produces the three fields of every level; the backward
 * launch returns goff = ((g3 m3 + g2) m2 + g1) m1 -- every product and sum a separately rounded fp32 operation, i.e.
 * bit-identical to the sequence of ATen multiplications / autograd additions it replaces.
 * Tensors: channels-last (B, C, H, W) with C = 2 * taps; `off` may have any image pitch (a slice of the concatenated
 * levels), everything else is dense.  forward reads off / writes out[3]; backward reads gout[6] (NULL = zero; gout[k] + gout[3 + k] is field k's gradient, as
 * autograd would have accumulated it for a field with two consumers) / writes goff. */
typedef struct {
    const float *off;
    float *out[3];
    const float *gout[6];   /* [0..2]: gradients of the three fields; [3..5]: of a second consumer's aliases of them (NULL = none) */
    float *goff;
    int64_t images, per_image, off_image_pitch;   /* B, H*W*C, floats between the images of `off` */
    float mh[3], mw[3];
} lsn_offset_chain_level;
int lsn_offset_chain_forward(int n_levels, const lsn_offset_chain_level *levels, int C, lsn_stream_t stream);
int lsn_offset_chain_backward(int n_levels, const lsn_offset_chain_level *levels, int C, lsn_stream_t stream);

/* ---- k nearest per column: centroid_assigner.py:74, atss_assigner.py:103-111 ----------------- */
/* x: row-major (P, G) matrix on the device, row pitch ldx floats (the assigners' points x gts distance matrix).  For
 * every column g and every row segment s = [seg_start[s], seg_start[s] + seg_len[s]) (host arrays, 1 <= nseg <= 8,
 * seg_len >= k) the k smallest (largest != 0: largest) entries, in that order, equal values by ascending row:
 *   values[(s * k + r) * G + g], indices[(s * k + r) * G + g] (row index in [0, P), int64)
 * -- `torch.topk(x[start:start + n], k, dim=0, largest=False)` of every segment in one launch (the reference calls it
 * once per FPN level and image; NaN orders as the largest value, as in torch).  [fused: replaces ATen's topk] */
int lsn_topk_columns(const float *x, int P, int G, int ldx, int nseg, const int *seg_start, const int *seg_len, int k,
                     int largest, float *values, int64_t *indices, lsn_stream_t stream);

/* ---- NMS: nms_ext.cpp:18-29 ---------------------------------------------------------------- */
/* dets (n,5) = x1,y1,x2,y2,score on the device.  keep (capacity n, int64, device) receives the
 * kept indices into the INPUT order, in descending score order; *num_keep (device int64) their
 * count.  IoU uses area=(x2-x1)*(y2-y1), suppress when IoU > thr (strict) -- nms_cpu.cpp:21-63,
 * nms_kernel.cu:14-22.  `order` (n, int64, device) must hold argsort(score, descending) computed
 * by the caller (the reference calls tensor.sort, nms_kernel.cu:81-83); `workspace` must hold
 * lsn_nms_workspace_bytes(n) bytes.  Fully asynchronous: no host sweep, no D2H copy. */
int64_t lsn_nms_workspace_bytes(int n);
int lsn_nms(const float *dets, const int64_t *order, int n, float iou_thr, int64_t *keep,
            int64_t *num_keep, void *workspace, lsn_stream_t stream);

/* ---- dense convolution (groups = 1), channels-last ----------------------------------------------
 * The reference runs torch.nn.Conv2d (cuDNN) for every dense conv of the path (resnet.py:624-631,261-301,
 * fpn.py:171-217, lsnet_head.py:160-257).  x (B,H,W,C), w (Co,kh,kw,C) = the channels-last image of the
 * (Co,C,kh,kw) weight, out (B,Ho,Wo,Co), fp32, 16-byte aligned; cross-correlation with zero padding like
 * F.conv2d.  Arithmetic: split-bf16 products with fp32 accumulation -- LSN_MATH_BF16X6 (fp32-equivalent) unless the
 * mode is LSN_MATH_BF16X3; there is no fp32-MFMA variant of these kernels (in LSN_MATH_FP32 the Python mirror keeps
 * the vendor library).  `relu` fuses max(., 0).  Supported: C % 4 == 0; tensors < 2 GiB.
 *
 * The kernels read the weight as a PREPARED IMAGE: bf16 planes in MFMA fragment order, which a wave fetches straight
 * into registers (csrc/conv_kernels.h).  kind 0 = forward image, kind 1 = backward-data image (transposed, taps flipped,
 * one sub-image per residue class of a strided convolution).  A caller that keeps the image rebuilds it only when the
 * weight changes (once per optimizer step; the Python mirror keys it on the parameter's version counter):
 *   lsn_conv2d_prepared_bytes    size of the image in bytes (< 0: unsupported stride / dilation combination)
 *   lsn_conv2d_prepare_weights   w -> image (depends on the math mode current at the call); _multi: many at once
 *   lsn_conv2d_forward_prepared / lsn_conv2d_backward_data_prepared   the passes proper; `xpitch` = C except for the
 *                                row-merged form below
 * The one-shot entry points below build the image inside the call, into `workspace` / `wt_workspace` (at least
 * lsn_conv2d_prepared_bytes bytes) or, when that is NULL, into a stream-ordered temporary.
 * backward_data: any stride -- every residue class (y mod stride, x mod stride) of input pixels is computed as its own
 * stride-1 convolution of grad_out over the taps that reach it; needs Co % 4 == 0 (pad grad_out and w with zero filters).
 * forward_pitched: the row-merged form for shallow inputs (the 7x7 stem on 3(+1) channels): `xpitch` floats separate
 * adjacent pixels while a tap spans C = n * xpitch consecutive floats (n pixels of the same row), kw = 1, pad = 0;
 * w is (Co, kh, 1, C) = the memory image of a (Co, kh, n, xpitch) weight.  The caller pads the image spatially. */
/* Batched forms: up to 16 input maps of different sizes that share one weight (the FPN levels under LSHead's shared
 * convolutions, lsnet_head.py:502-513) in ONE launch each way.  forward: x -> out.  backward_data: x = grad_out
 * (B,Ho,Wo,Co), out = grad_in (B,H,W,C), and B/H/W are the forward INPUT sizes; stride 1 when n_levels > 1.
 * backward_weight: x = forward input, grad_out; the weight / bias gradients are summed over the levels. */
typedef struct lsn_conv_level {
    const float *x;
    float *out;
    const float *grad_out;   /* backward_weight only */
    int B, H, W;
    const float *residual;   /* forward / backward_data, optional: a tensor of the output's shape ADDED before the ReLU --
                              * the `relu(bn(conv(x)) + identity)` tail of a ResNet block whose BatchNorm has been folded
                              * into the weight and bias (resnet.py:261-301); in backward_data the gradient that reaches
                              * the same tensor along another path (the identity branch).  May alias `out` (accumulate) */
    const float *gate;       /* forward / backward_data, optional: a tensor of the output's shape; the finished element
                              * (after bias, residual, ReLU) is set to 0 where gate <= 0 -- the ReLU gate of the activation
                              * whose gradient a backward_data launch produces, applied in its epilogue instead of by a
                              * pass of its own.  backward_data with stride > 1 takes residual / gate only when every
                              * residue class of input pixels has a tap (3x3 stride 2: yes; 1x1 stride 2: no) */
} lsn_conv_level;
int64_t lsn_conv2d_prepared_bytes(int kind, int C, int Co, int kh, int kw, int stride, int pad, int dil);
int lsn_conv2d_prepare_weights(int kind, const float *w, void *prepared, int C, int Co, int kh, int kw, int stride,
                               int pad, int dil, lsn_stream_t stream);
/* The images of many weights in ONE launch (every trainable convolution after an optimizer step). */
typedef struct lsn_conv_wprep {
    int kind;            /* 0 forward, 1 backward-data */
    const float *w;
    void *prepared;
    int C, Co, kh, kw, stride, pad, dil;
    /* Optional (all NULL otherwise): an eval-mode BatchNorm behind the convolution, folded into the image.
     * Every weight of (forward) output channel co is multiplied by a[co] = bn_gamma[co] / sqrt(bn_var[co] + bn_eps), and
     * (kind 0) shift_out[co] = bn_beta[co] - bn_mean[co] * a[co] (Co floats) is written for the `bias` argument of
     * lsn_conv2d_forward_prepared: relu(bn(conv(x)) + residual) of a ResNet block (resnet.py:261-301) is then ONE launch,
     * with the affine parameters still trainable.  kind 1 (shift_out NULL): the backward-data image of the SCALED weight,
     * so that lsn_conv2d_backward_data_prepared takes the gated upstream gradient as it is; the parameter gradients come
     * from lsn_conv2d_backward_weight_bn. */
    const float *bn_gamma, *bn_var, *bn_beta, *bn_mean;
    float *shift_out;
    float bn_eps;
} lsn_conv_wprep;
int lsn_conv2d_prepare_weights_multi(int n_items, const lsn_conv_wprep *items, lsn_stream_t stream);
/* One item, launched directly (no job table: usable while a hipGraph is being captured). */
int lsn_conv2d_prepare_weights_item(const lsn_conv_wprep *item, lsn_stream_t stream);
int lsn_conv2d_forward_prepared(int n_levels, const lsn_conv_level *levels, const void *prepared, const float *bias,
                                int C, int xpitch, int Co, int kh, int kw, int stride, int pad, int dil, int relu,
                                lsn_stream_t stream);
int lsn_conv2d_backward_data_prepared(int n_levels, const lsn_conv_level *levels, const void *prepared, int C,
                                      int Co, int kh, int kw, int stride, int pad, int dil, lsn_stream_t stream);
int lsn_conv2d_forward(const float *x, const float *w, const float *bias, float *out, void *workspace, int B, int H,
                       int W, int C, int Co, int kh, int kw, int stride, int pad, int dil, int relu,
                       lsn_stream_t stream);
int lsn_conv2d_forward_multi(int n_levels, const lsn_conv_level *levels, const float *w, const float *bias, void *workspace,
                             int C, int Co, int kh, int kw, int stride, int pad, int dil, int relu, lsn_stream_t stream);
int lsn_conv2d_backward_data_multi(int n_levels, const lsn_conv_level *levels, const float *w, float *wt_workspace, int C,
                                   int Co, int kh, int kw, int stride, int pad, int dil, lsn_stream_t stream);
int lsn_conv2d_backward_weight_multi(int n_levels, const lsn_conv_level *levels, float *grad_w, float *grad_bias, int C,
                                     int Co, int kh, int kw, int stride, int pad, int dil, int accumulate,
                                     lsn_stream_t stream);
int lsn_conv2d_forward_pitched(const float *x, const float *w, const float *bias, float *out, void *workspace, int B,
                               int H, int W, int C, int xpitch, int Co, int kh, int kw, int stride, int pad, int dil,
                               int relu, lsn_stream_t stream);
int lsn_conv2d_backward_data(const float *grad_out, const float *w, float *grad_in, float *wt_workspace, int B,
                             int H, int W, int C, int Co, int kh, int kw, int stride, int pad, int dil,
                             lsn_stream_t stream);
/* grad_w (Co,kh,kw,C) and optionally grad_bias (Co); any stride / padding / dilation.
 * `accumulate` (all parameter-gradient entry points of this header): 0 = the outputs are OVERWRITTEN; 1 = the results are
 * ADDED to what the buffers hold -- the caller keeps every parameter gradient of a step in one arena it zeroes once
 * (the all-reduce buckets of data-parallel training: no per-tensor memset, no gradient -> bucket copy or add). */
int lsn_conv2d_backward_weight(const float *x, const float *grad_out, float *grad_w, float *grad_bias, int B, int H,
                               int W, int C, int Co, int kh, int kw, int stride, int pad, int dil, int accumulate,
                               lsn_stream_t stream);
/* Parameter gradients of y = bn_eval(conv(x, w)) with the BatchNorm folded into the convolution (lsn_conv_wprep): the
 * reference runs aten::convolution_backward and the backward of F.batch_norm in eval mode (resnet.py:261-301).  g = the
 * upstream gradient w.r.t. the NORMALISED output (after the ReLU gate), a_c = gamma_c / sqrt(var_c + eps):
 *   G[co][k][c]     = sum_p g[p][co] x[p @ k][c]                      (the weight gradient of the raw convolution under g)
 *   grad_w          (+)= a_co G[co]
 *   grad_beta[co]   (+)= sum_p g[p][co]
 *   grad_gamma[co]  (+)= (sum_{k,c} w[co][k][c] G[co][k][c] - mean_co sum_p g[p][co]) / sqrt(var_co + eps)
 * -- sum_p g (conv - mean) rstd with conv = w . x pulled out of the pixel sum: exact for EVERY gamma, zero included
 * (zero_init_residual), where recovering x_hat from the stored activation as (y - residual - beta) / gamma cannot be.  One
 * weight-gradient launch plus its ordered reduce; bit-identical run to run. */
int lsn_conv2d_backward_weight_bn(const float *x, const float *g, const float *w, const float *bn_gamma,
                                  const float *bn_mean, const float *bn_var, float bn_eps, float *grad_w,
                                  float *grad_gamma, float *grad_beta, int B, int H, int W, int C, int Co, int kh, int kw,
                                  int stride, int pad, int dil, int accumulate, lsn_stream_t stream);

/* The same for up to 8 convolutions of ONE geometry (B, H, W, C, Co, kernel, stride, pad, dil) with their own tensors --
 * the identical bottlenecks of a ResNet stage (resnet.py:ResLayer builds blocks 1 .. n-1 alike): one launch instead of
 * n_jobs small ones.  A stage-3 1x1 weight gradient alone has 16 output tiles and needs ~32 pixel splits to fill the
 * chip; five of them together need ~6, with a fifth of the partial-tile traffic per job.  Results as from n_jobs calls of
 * lsn_conv2d_backward_weight_bn up to the summation order over pixels. */
typedef struct {
    const float *x, *g, *w, *bn_gamma, *bn_mean, *bn_var;
    float bn_eps;
    float *grad_w, *grad_gamma, *grad_beta;
} lsn_wgrad_bn_job;
int lsn_conv2d_backward_weight_bn_jobs(int n_jobs, const lsn_wgrad_bn_job *jobs, int B, int H, int W, int C, int Co, int kh,
                                       int kw, int stride, int pad, int dil, int accumulate, lsn_stream_t stream);

/* ---- Grouped convolution (ResNeXt bottlenecks) -------------------------------------------------
 * Reference: torch.nn.Conv2d(groups = G) as built by mmdet/models/backbones/resnext.py:11-83 (Bottleneck.conv2:
 * 3x3, G = 64, width / G = 4 .. 32 channels per group), i.e. ATen's grouped convolution and its backward
 * (aten::convolution_backward).  NHWC activations, OHWI weights w[co][i][j][ci] with ci in [0, C / G).
 * Exact fp32 arithmetic (fmaf chains on the vector ALU; the products are far too thin for the matrix pipe and the
 * layer is bound by HBM), in every math mode.  Supported: C / G == Co / G in {4, 8, 16, 32}, kh * kw <= 9, any
 * stride / padding / dilation; anything else returns LSN_ERR_UNSUPPORTED (the caller keeps ATen).
 * forward: out (B,Ho,Wo,Co) = conv(x (B,H,W,C)) + bias, optional ReLU.  backward_data: grad_in (B,H,W,C), OVERWRITTEN.
 * backward_weight: grad_w (Co,kh,kw,C/G) and optionally grad_bias (Co), OVERWRITTEN or (accumulate = 1) added to. */
int lsn_grouped_conv2d_forward(const float *x, const float *w, const float *bias, float *out, int B, int H, int W, int C,
                               int Co, int kh, int kw, int stride, int pad, int dil, int groups, int relu,
                               lsn_stream_t stream);
int lsn_grouped_conv2d_backward_data(const float *grad_out, const float *w, float *grad_in, int B, int H, int W, int C,
                                     int Co, int kh, int kw, int stride, int pad, int dil, int groups,
                                     lsn_stream_t stream);
int lsn_grouped_conv2d_backward_weight(const float *x, const float *grad_out, float *grad_w, float *grad_bias, int B,
                                       int H, int W, int C, int Co, int kh, int kw, int stride, int pad, int dil,
                                       int groups, int accumulate, lsn_stream_t stream);

/* ---- GroupNorm (+ReLU) on channels-last tensors ----------------------------------------------
 * The reference uses torch.nn.GroupNorm followed by nn.ReLU (ATen kernels; call sites
 * lsnet_head.py:1830-1849,136-141 and ConvModule in fpn.py:65-156).  These entry points are the fused
 * channels-last equivalent: y = relu?( (x - mean_g) * rstd_g * gamma_c + beta_c ), statistics per
 * (image, group) over HW x C/G elements, biased variance, eps inside the square root -- the ATen definition.
 * Several tensors that share one GroupNorm module (the FPN levels) are processed per call.
 * x / y / dy / dx: (B, HW, C) channels-last fp32, 16-byte aligned.  Supported: C % 4 == 0, C/G a multiple of 4 or one
 * of 1, 2 (round 6), 256 % (C/4) == 0 (C = 4 ... 1024); otherwise LSN_ERR_UNSUPPORTED (the caller keeps ATen's GroupNorm).
 * mean_rstd: (sum of B over levels, G, 2) fp32, written by forward and read by backward.
 * workspace: lsn_group_norm_workspace_bytes() bytes of device memory, contents undefined.
 * backward recomputes the ReLU gate from x (y is not needed); grad_gamma / grad_beta (C) are OVERWRITTEN with
 * the sums over all levels and images and may be NULL. */
typedef struct lsn_gn_level {
    const float *x;    /* input                              */
    float *y;          /* forward: output                    */
    const float *dy;   /* backward: gradient w.r.t. output   */
    float *dx;         /* backward: gradient w.r.t. input    */
    int B, HW;
    /* floats between the images of y / dy; 0 = dense (HW * C).  A level of the concatenated pixel tensor LSHead runs its
     * pointwise convolutions on ((B, N_all, C): the pixel rows of all levels back to back, lsnet_head.py:640-755) is a view
     * with the batch stride N_all * C: the forward writes there, the backward reads the gradient there -- no torch.cat in
     * front of the 1x1 convolution, no per-level copies of its input gradient behind it. */
    long long y_batch_stride, dy_batch_stride;
} lsn_gn_level;
int64_t lsn_group_norm_workspace_bytes(int n_levels, const lsn_gn_level *levels, int C, int G);
int lsn_group_norm_forward(int n_levels, const lsn_gn_level *levels, int C, int G, const float *gamma,
                           const float *beta, float eps, int relu, float *mean_rstd, void *workspace,
                           lsn_stream_t stream);
int lsn_group_norm_backward(int n_levels, const lsn_gn_level *levels, int C, int G, const float *gamma,
                            const float *beta, int relu, const float *mean_rstd, float *grad_gamma,
                            float *grad_beta, void *workspace, int accumulate, lsn_stream_t stream);

/* ---- BatchNorm with frozen statistics (+ residual add, + ReLU), channels-last ------------------
 * The LSNet backbones keep every BatchNorm in eval mode while training (norm_eval=True, resnet.py:636-645) but
 * train its affine parameters; the reference runs F.batch_norm, the residual add and F.relu as separate ATen /
 * cuDNN kernels (resnet.py:261-301).  Fused equivalents on (N = B*H*W, C) channels-last fp32 tensors:
 *   forward : y = act( (x - mean_c) / sqrt(var_c + eps) * gamma_c + beta_c (+ residual) )
 *   backward: dz = grad_y * [y > 0] (when relu);  grad_x = dz * gamma_c / sqrt(var_c + eps);  grad_residual = dz;
 *             grad_gamma_c = sum dz * (x - mean_c) / sqrt(var_c + eps);  grad_beta_c = sum dz
 * residual / grad_x / grad_residual / grad_gamma+grad_beta may be NULL (not needed).  grad_gamma / grad_beta are
 * OVERWRITTEN, or added to when accumulate = 1 (see lsn_conv2d_backward_weight).  Supported: C % 4 == 0 and 256 % (C/4) == 0 (C = 4 ... 1024), or C a multiple of 1024 (the 2048-channel
 * maps of a ResNet's last stage); else LSN_ERR_UNSUPPORTED. */
int lsn_bn_eval_act_forward(const float *x, const float *residual, float *y, const float *running_mean,
                            const float *running_var, const float *gamma, const float *beta, float eps, int relu,
                            int N, int C, lsn_stream_t stream);
int64_t lsn_bn_eval_act_workspace_bytes(int N, int C);   /* scratch for backward when grad_gamma is wanted */
int lsn_bn_eval_act_backward(const float *grad_y, const float *y, const float *x, const float *running_mean,
                             const float *running_var, const float *gamma, float eps, int relu, float *grad_x,
                             float *grad_residual, float *grad_gamma, float *grad_beta, void *workspace, int N,
                             int C, int accumulate, lsn_stream_t stream);
/* grad[i] = y[i] > 0 ? grad_y[i] : 0 over n floats (n % 4 == 0, 16-byte aligned): the gradient through a ReLU whose
 * OUTPUT y was stored (F.relu's backward, resnet.py:261-301).  The only stand-alone pass left of the backward of a
 * conv + eval-BatchNorm (+ residual) + ReLU block whose norm is folded into the convolution (lsn_conv_wprep): the scale
 * a_c rides in the backward-data image, the parameter gradients come from lsn_conv2d_backward_weight_bn, and where the
 * gradient is produced by one of this library's backward-data launches the gate rides in its epilogue instead
 * (lsn_conv_level.gate). */
int lsn_relu_gate(const float *grad_y, const float *y, float *grad, int64_t n, lsn_stream_t stream);
/* The same for up to 8 tensors in one launch -- the FPN levels of one multi-level convolution with a ReLU epilogue
 * (lsn_conv2d_forward_multi: LSHead's init and fusion convolutions, lsnet_head.py:502-638).  y and grad: B dense images of
 * per_image floats; grad_y: the same images gy_batch_stride floats apart (per_image when dense) -- the gradient of a level that
 * is a slice of the head's concatenated pixel tensor is gated where it lies, without a copy.  per_image, gy_batch_stride: multiples
 * of 4; 16-byte aligned pointers. */
typedef struct lsn_gate_job {
    const float *grad_y, *y;
    float *grad;
    int B;
    int64_t per_image, gy_batch_stride;
} lsn_gate_job;
int lsn_relu_gate_multi(int n_jobs, const lsn_gate_job *jobs, lsn_stream_t stream);

/* ---- diagnostics ---------------------------------------------------------------------------- */
/* When set to a device buffer of 512 int64 (NULL disables), thread 0 of workgroup `block` of the
 * next DCN forward / backward-data launches appends (phase_id << 56 | shader_clock) stamps at its
 * phase boundaries: a per-chunk cycle anatomy for tuning.  Not thread-safe; profiling only. */
int lsn_debug_phase_clocks(long long *device_buf_512, int block);

/* Per-kernel-family launch timing.  lsn_prof_enable(1) clears the log and makes every launch of the instrumented
 * families record a HIP event pair on its launch stream; lsn_prof_read() waits for the recorded events and returns one
 * entry per family -- dcn_fwd, dcn_bwd_data, dcn_wgrad, conv_fwd, conv_bwd_data, conv_wgrad, norm, gconv -- with the
 * launch count, the summed kernel time and the summed ALGORITHMIC flops / bytes of those launches (contractions: 2 x
 * output pixels x Co x C/groups x kh x kw flops; bytes: each operand read or written once).
 * lsn_prof_enable(on): 0 = off, 1 = every family, any other value = a bit mask of families in the order above (bit 0 =
 * dcn_fwd ...), so that a timed run can carry the events of ONE family only (a few dozen launches per step) after a
 * warm-up run with all of them found out which one dominates.
 * Returns the number of entries written (8) or a negative lsn error.  Not thread-safe. */
typedef struct lsn_prof_entry {
    char name[48];
    long long launches;
    double total_ms;
    double flops;
    double bytes;
} lsn_prof_entry;
int lsn_prof_enable(int on);
int lsn_prof_read(lsn_prof_entry *out, int max_entries);

/* Per-call log of the dense convolutions (forward / backward-data entry points, prepared or one-shot): while it is on, every
 * call records its arguments and a pair of HIP events on the launch stream.  lsn_prof_launch_log(on) clears the log and
 * switches it; lsn_prof_read_launches(NULL, 0) returns the number of records, with a buffer it waits for the events and
 * copies up to max_entries records in call order.  tools/instep_vs_isolated.py replays every record back to back on its own
 * and compares (profiles/r6_instep_vs_isolated.txt).  kind 0: forward, 1: backward-data (B/H/W = the forward INPUT sizes). */
typedef struct lsn_prof_launch {
    int kind, C, Co, kh, kw, stride, pad, dil, relu, xpitch, n_levels, has_residual, has_gate;
    int B[16], H[16], W[16];
    float ms;
} lsn_prof_launch;
int lsn_prof_launch_log(int on);
int lsn_prof_read_launches(lsn_prof_launch *out, int max_entries);

/* What the library itself has asked of the HIP runtime outside kernel launches since it was loaded:
 *   out4[0] hipMalloc calls (library-owned scratch: partial tiles, stream-K slots and counters, tables -- grown on demand,
 *           never freed), out4[1] bytes they hold, out4[2] blocking stream synchronisations (a weight-image job table
 *           that changed), out4[3] stream-ordered pool allocations (hipMallocAsync / hipFreeAsync: no synchronisation).
 * A training loop whose input shapes change every iteration (the reference's multi-scale training,
 * configs/lsnet/lsnet_bbox_r50_fpn_mstrain_2x_coco.py:13-15) must reach a state in which out4[0] and out4[2] stay put. */
int lsn_scratch_stats(long long *out4);

/* Deferred weight-gradient reduces.  Every weight-gradient entry point of this header (dense, folded-norm, jobs, deformable)
 * ends in a reduce launch over its partial tiles.  lsn_wgrad_defer(max_mbytes > 0, stream): from now on the calls on `stream`
 * that ACCUMULATE into their gradient (accumulate = 1: a gradient arena owns the memory and nobody reads it before the step's
 * gradients are complete) leave their partial tiles in a library arena and queue the reduce; the queue is run -- one launch per
 * reduce kind for up to 16 gradients, the same arithmetic per element in the same order -- by lsn_wgrad_flush(stream), by a
 * second call for a gradient that already has a queued one, whenever the arena holds more than max_mbytes, and by
 * lsn_wgrad_defer(0, stream), which also ends the mode.  Calls with accumulate = 0 are never deferred.  A caller that enables
 * the mode owes a flush before anything reads the gradients (lsnet_amd/parallel/reducer.py: before a bucket's all-reduce, in
 * finish()). */
int lsn_wgrad_defer(int max_mbytes, lsn_stream_t stream);
int lsn_wgrad_flush(lsn_stream_t stream);

/* D = A(MxK) * B(KxN) through the same MFMA fragment code as the DCN kernels (self-test). */
int lsn_selftest_mfma(const float *A, const float *B, float *D, int M, int N, int K, int variant,
                      lsn_stream_t stream);

/* ---- input preparation (csrc/image.hip) --------------------------------------------------------------------
 * The reference prepares images on the CPU: Resize / RandomFlip / Normalize / Pad of
 * mmdet/datasets/pipelines/transforms.py:184-207, 409-432, 484-495, 550-565 and the zero-padding of
 * mmcv/parallel/collate.py:34-64.  One launch does the same for one uploaded 8-bit image:
 *   src   device pointer, h x w x c interleaved uint8 as decoded (c <= 4);
 *   dh,dw size after the resize (cv2.INTER_LINEAR rule, bit-identical to the host implementation);
 *   flip  mirror the RESIZED image horizontally / vertically;  reverse_channels: BGR -> RGB;
 *   mean, inv_std  host pointers, c floats each, in OUTPUT channel order;
 *   dst   device pointer to this image's slot: out_h x out_w x c float32 (channels-last), everything outside
 *         dh x dw is set to pad_val. */
int lsn_image_prep_u8(const uint8_t *src, int sh, int sw, int c, int dh, int dw, int flip_h, int flip_v,
                      const float *mean, const float *inv_std, int reverse_channels, float pad_val, float *dst,
                      int out_h, int out_w, lsn_stream_t stream);

/* ---- fused cross-IOU loss of the bbox task (csrc/loss.hip) ---------------------------------------------------
 * mmdet/models/losses/cross_iou_loss.py:10-33, 61-132 with loss_type='bbox' for n points in one launch:
 *   pred, target  (n, 20) float32, 16-byte aligned rows: 5 landmarks x [y_up, y_down, x_left, x_right];
 *   active        (n, 20) bytes: non-zero where a component carries the ground truth (the reference's `pos_inds`);
 *   anchor (n, 2), bbox_gt (n, 4); weight (n) or NULL;
 * forward : loss_rows[i] = weight[i] * loss_i  (reduction 'none');
 * backward: grad_pred[i, :] = grad_rows[i] * weight[i] * d loss_i / d pred[i, :]. */
int lsn_cross_iou_bbox_forward(const float *pred, const float *target, const uint8_t *active, const float *anchor,
                               const float *bbox_gt, const float *weight, int64_t n, float alpha, float eps,
                               float *loss_rows, lsn_stream_t stream);
int lsn_cross_iou_bbox_backward(const float *pred, const float *target, const uint8_t *active, const float *anchor,
                                const float *bbox_gt, const float *weight, const float *grad_rows, int64_t n,
                                float alpha, float eps, float *grad_pred, lsn_stream_t stream);

/* The whole bbox regression stage of LSHead.loss_single for n points (lsnet_head.py:402-427, 1066-1101): pred_raw (n, 20)
 * in stride units; gt_pts (n, 10) extreme points + centre as (x, y); anchor3 (n, 3) = (x, y, stride); bbox_gt (n, 4);
 * weight (n): 0 for points without an object.  Scaling to pixels, normalisation by base_scale * stride, target and
 * active-half construction and the cross-IOU row happen in registers. */
int lsn_cross_iou_bbox_stage_forward(const float *pred_raw, const float *gt_pts, const float *anchor3, const float *bbox_gt,
                                     const float *weight, int64_t n, float base_scale, float alpha, float eps,
                                     float *loss_rows, lsn_stream_t stream);
int lsn_cross_iou_bbox_stage_backward(const float *pred_raw, const float *gt_pts, const float *anchor3, const float *bbox_gt,
                                      const float *weight, const float *grad_rows, int64_t n, float base_scale, float alpha,
                                      float eps, float *grad_raw, lsn_stream_t stream);

/* The polygon (instance segmentation: nv contour vectors + centre) and keypoint (pose: nv keypoints + centre) variants
 * of the same loss -- mmdet/models/losses/cross_iou_loss.py:68-77 and :80-94, call sites lsnet_head.py:1103-1270 --
 * for n points in one launch each way.  Rows have ncomp = 4 (nv + 1) components (148 for 36 contour vectors, 72 for 17
 * keypoints).  kind 1 = polygon: overlap = mean over `sub` interleaved landmark subsets (cross_iou_loss.py: stride = 9)
 * of sum(min) / sum(max), plus the distance / aspect terms on the bounding box of the nv vectors; needs anchor (n, 2)
 * and bbox_gt (n, 4).  kind 2 = keypoint: mean over the (neg, pos) pairs of sum(min) / sum(max clamped at eps), a
 * keypoint's two pairs counted when vs (n, nv) > 0, the centre pairs always; anchor / bbox_gt unused (may be NULL).
 * weight (n) may be NULL.  forward: loss_rows (n) = weight x row loss.  backward: grad_pred (n, ncomp), OVERWRITTEN,
 * = grad_rows x weight x d row / d pred (ties of max / min split 0.5 / 0.5 as ATen does). */
int lsn_cross_iou_rows_forward(int kind, const float *pred, const float *target, const uint8_t *active, const float *anchor,
                               const float *bbox_gt, const float *vs, const float *weight, int64_t n, int ncomp, int sub,
                               float alpha, float eps, float *loss_rows, lsn_stream_t stream);
int lsn_cross_iou_rows_backward(int kind, const float *pred, const float *target, const uint8_t *active, const float *anchor,
                                const float *bbox_gt, const float *vs, const float *weight, const float *grad_rows, int64_t n,
                                int ncomp, int sub, float alpha, float eps, float *grad_pred, lsn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LSNET_HIP_H_ */
