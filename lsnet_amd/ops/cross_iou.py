"""Fused cross-IOU loss (csrc/loss.hip): the per-point loss and its gradient in one launch each, instead of the ~60
elementwise launches of the torch formulation (models/losses/cross_iou_loss.py) -- the bbox task
(lsn_cross_iou_bbox_forward / _backward and the whole-stage variant), the polygon (instance segmentation) and the
keypoint (pose) tasks (lsn_cross_iou_rows_forward / _backward).  On by default for device tensors (verified on the MI355X against the torch
formulation and the reference fixture: tests/test_zz_fused_ciou_gpu.py); `LSNET_FUSED_CIOU=0` keeps the torch formulation."""
import ctypes
import os

import torch

from .. import _lib


def enabled():
    return os.environ.get('LSNET_FUSED_CIOU', '1') != '0'


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _CrossIouBbox(torch.autograd.Function):

    @staticmethod
    def forward(ctx, pred, target, active, anchor, bbox_gt, weight, alpha, eps):
        pred, target = pred.contiguous(), target.contiguous()
        active = active.to(torch.uint8).contiguous()
        anchor, bbox_gt = anchor.contiguous(), bbox_gt.contiguous()
        weight = None if weight is None else weight.contiguous()
        loss = torch.empty(pred.shape[0], dtype=torch.float32, device=pred.device)
        _lib.check(_lib.load().lsn_cross_iou_bbox_forward(_p(pred), _p(target), _p(active), _p(anchor), _p(bbox_gt), _p(weight),
                                                          ctypes.c_int64(pred.shape[0]), ctypes.c_float(alpha),
                                                          ctypes.c_float(eps), _p(loss), _stream()))
        ctx.save_for_backward(pred, target, active, anchor, bbox_gt, *([weight] if weight is not None else []))
        ctx.cfg = (alpha, eps, weight is not None)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_rows):
        alpha, eps, has_w = ctx.cfg
        pred, target, active, anchor, bbox_gt, *rest = ctx.saved_tensors
        weight = rest[0] if has_w else None
        grad = torch.empty_like(pred)
        _lib.check(_lib.load().lsn_cross_iou_bbox_backward(_p(pred), _p(target), _p(active), _p(anchor), _p(bbox_gt), _p(weight),
                                                           _p(grad_rows.contiguous()), ctypes.c_int64(pred.shape[0]),
                                                           ctypes.c_float(alpha), ctypes.c_float(eps), _p(grad), _stream()))
        return grad, None, None, None, None, None, None, None


def usable(pred, target, loss_type):
    return (loss_type == 'bbox' and pred.is_cuda and pred.dtype == torch.float32 and pred.dim() == 2
            and pred.shape[1] == 20 and not target.requires_grad)


_KIND = {'polygon': 1, 'keypoint': 2}


def rows_usable(pred, target, loss_type, stride=9, active=None, anchor=None, bbox_gt=None, vs=None, weight=None):
    """The polygon / keypoint kernels: rows of 4 (nv + 1) fp32 components on the device.  The kernels index the side
    tensors unchecked (active[M i], anchor[2 i], bbox_gt[4 i], vs[(M / 4 - 1) i], weight[i]): every tensor handed in is
    verified here -- shape, device -- and a mismatch sends the caller to the torch formulation, which raises the shape
    error a user expects instead of reading out of bounds."""
    if not (loss_type in _KIND and pred.is_cuda and pred.dtype == torch.float32 and pred.dim() == 2
            and pred.shape[1] >= 8 and pred.shape[1] % 4 == 0 and not target.requires_grad
            and (loss_type == 'keypoint' or 1 <= stride <= min(16, pred.shape[1] // 4))):
        return False
    n, m = pred.shape
    want = ((target, (n, m)), (active, (n, m)), (anchor, (n, 2)), (bbox_gt, (n, 4)), (vs, (n, m // 4 - 1)), (weight, (n,)))
    for t, shape in want:
        if t is not None and (tuple(t.shape) != shape or t.device != pred.device):
            return False
    return True


class _CrossIouRows(torch.autograd.Function):

    @staticmethod
    def forward(ctx, pred, target, active, anchor, bbox_gt, vs, weight, kind, sub, alpha, eps):
        c = lambda t: None if t is None else t.contiguous()
        pred, target, anchor, bbox_gt, weight = c(pred), c(target), c(anchor), c(bbox_gt), c(weight)
        vs = None if vs is None else vs.to(torch.float32).contiguous()
        active = active.to(torch.uint8).contiguous()
        n, m = pred.shape
        loss = torch.empty(n, dtype=torch.float32, device=pred.device)
        _lib.check(_lib.load().lsn_cross_iou_rows_forward(kind, _p(pred), _p(target), _p(active), _p(anchor), _p(bbox_gt), _p(vs),
                                                          _p(weight), ctypes.c_int64(n), m, sub, ctypes.c_float(alpha),
                                                          ctypes.c_float(eps), _p(loss), _stream()))
        ctx.save_for_backward(pred, target, active, anchor, bbox_gt, vs, weight)   # (None entries are allowed)
        ctx.cfg = (kind, sub, alpha, eps)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_rows):
        kind, sub, alpha, eps = ctx.cfg
        pred, target, active, anchor, bbox_gt, vs, weight = ctx.saved_tensors
        n, m = pred.shape
        grad = torch.empty_like(pred)
        _lib.check(_lib.load().lsn_cross_iou_rows_backward(kind, _p(pred), _p(target), _p(active), _p(anchor), _p(bbox_gt), _p(vs),
                                                           _p(weight), _p(grad_rows.contiguous()), ctypes.c_int64(n), m, sub,
                                                           ctypes.c_float(alpha), ctypes.c_float(eps), _p(grad), _stream()))
        return (grad,) + (None,) * 10


def cross_iou_rows(pred, target, active, loss_type, anchor=None, bbox_gt=None, vs=None, weight=None, alpha=0.2, eps=1e-6,
                   stride=9):
    """(n,) weighted loss rows (reduction 'none') of the polygon / keypoint cross-IOU loss; gradient flows to `pred` only
    (the other inputs are targets; they are kept alive for the backward launch)."""
    return _CrossIouRows.apply(pred, target, active, anchor, bbox_gt, vs, weight, _KIND[loss_type], int(stride), float(alpha),
                               float(eps))


def cross_iou_bbox_rows(pred, target, active, anchor, bbox_gt, weight=None, alpha=0.2, eps=1e-6):
    """(n,) weighted loss rows (reduction 'none') of the bbox cross-IOU loss; gradient flows to `pred` only."""
    return _CrossIouBbox.apply(pred, target, active, anchor, bbox_gt, weight, float(alpha), float(eps))


class _CrossIouBboxStage(torch.autograd.Function):

    @staticmethod
    def forward(ctx, raw, gt_pts, anchor3, bbox_gt, weight, base, alpha, eps):
        raw, gt_pts, anchor3 = raw.contiguous(), gt_pts.contiguous(), anchor3.contiguous()
        bbox_gt, weight = bbox_gt.contiguous(), weight.contiguous()
        loss = torch.empty(raw.shape[0], dtype=torch.float32, device=raw.device)
        _lib.check(_lib.load().lsn_cross_iou_bbox_stage_forward(_p(raw), _p(gt_pts), _p(anchor3), _p(bbox_gt), _p(weight),
                                                                ctypes.c_int64(raw.shape[0]), ctypes.c_float(base),
                                                                ctypes.c_float(alpha), ctypes.c_float(eps), _p(loss), _stream()))
        ctx.save_for_backward(raw, gt_pts, anchor3, bbox_gt, weight)
        ctx.cfg = (base, alpha, eps)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_rows):
        base, alpha, eps = ctx.cfg
        raw, gt_pts, anchor3, bbox_gt, weight = ctx.saved_tensors
        grad = torch.empty_like(raw)
        _lib.check(_lib.load().lsn_cross_iou_bbox_stage_backward(_p(raw), _p(gt_pts), _p(anchor3), _p(bbox_gt), _p(weight),
                                                                 _p(grad_rows.contiguous()), ctypes.c_int64(raw.shape[0]),
                                                                 ctypes.c_float(base), ctypes.c_float(alpha), ctypes.c_float(eps),
                                                                 _p(grad), _stream()))
        return grad, None, None, None, None, None, None, None


def cross_iou_bbox_stage_rows(pred_raw, gt_pts, anchor3, bbox_gt, weight, base_scale, alpha=0.2, eps=1e-6):
    """The bbox regression stage of LSHead for n points in one launch (see lsn_cross_iou_bbox_stage_forward): weighted
    loss rows; gradient flows to `pred_raw`."""
    return _CrossIouBboxStage.apply(pred_raw, gt_pts, anchor3, bbox_gt, weight, float(base_scale), float(alpha), float(eps))
