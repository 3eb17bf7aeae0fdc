"""Cross-IOU loss over landmark vectors (mmdet/models/losses/cross_iou_loss.py:10-172).

Every landmark offset is represented by four non-negative components [y_up, y_down, x_left,
x_right] of which one per (up,down)/(left,right) pair is "active" for the ground truth; the
inactive target component is set to alpha * active, and the overlap is sum(min(p,t)) / sum(max(p,t)).
bbox / polygon variants add a DIoU/CIoU-style penalty on the box spanned by the landmarks.

Written with `where` instead of the reference's masked in-place assignments: same values, no
data-dependent shapes, and no `torch.any(weight > 0)` host sync (an all-zero weight simply gives
a zero loss and zero gradient)."""
import math

import torch
import torch.nn as nn

from ..builder import LOSSES
from .utils import weighted_loss


def _signed(pairs):
    """(..., 2) non-negative (neg, pos) components -> signed scalar; index 0 wins ties and is
    negated, as `torch.max(..., dim)` + `inds == 0` does in the reference (:11-14, 37-40)."""
    neg, pos = pairs[..., 0], pairs[..., 1]
    return torch.where(pos > neg, pos, -neg)


def get_bbox_from_extreme(pred, anchor_pts):
    """pred (N,20) -> box (N,4) = [x of left lm, y of top lm, x of right lm, y of bottom lm] + anchor."""
    yx = _signed(pred.reshape(pred.shape[0], -1, 2)).reshape(pred.shape[0], -1, 2)   # (N,5,[y,x])
    x = yx[:, :, 1] + anchor_pts[:, 0:1]
    y = yx[:, :, 0] + anchor_pts[:, 1:2]
    return torch.stack([x[:, 1], y[:, 0], x[:, 3], y[:, 2]], dim=1)


def get_bbox_from_polygon(pred, anchor_pts):
    """pred (N,4*(nv+1)) -> bounding box of the nv polygon / keypoint vectors (centre excluded)."""
    body = pred[:, :-4]
    yx = _signed(body.reshape(body.shape[0], -1, 2)).reshape(body.shape[0], -1, 2)
    x = yx[:, :, 1] + anchor_pts[:, 0:1]
    y = yx[:, :, 0] + anchor_pts[:, 1:2]
    return torch.stack([x.min(1)[0], y.min(1)[0], x.max(1)[0], y.max(1)[0]], dim=1)


@weighted_loss
def cross_iou_loss(pred, target, loss_type=None, anchor_pts=None, vs=None, bbox_gt=None, pos_inds=None,
                   eps=1e-6, alpha=0.2, stride=9):
    # inactive component of every pair := alpha * active component  (:65-66)
    t2 = target.reshape(target.shape[0], -1, 2)
    p2 = pos_inds.reshape(target.shape[0], -1, 2)
    active = torch.where(p2[..., 0], t2[..., 0], t2[..., 1])
    target = torch.where(p2, t2, (alpha * active).unsqueeze(-1)).reshape(target.shape)

    l_max, l_min = torch.max(pred, target), torch.min(pred, target)
    if loss_type == 'polygon':
        # nine interleaved subsets of the landmarks (i, i+stride, ...), overlaps averaged (:68-77)
        n = pred.shape[0]
        mx, mn = l_max.reshape(n, -1, 4), l_min.reshape(n, -1, 4)
        parts = [mn[:, i::stride].reshape(n, -1).sum(1) / mx[:, i::stride].reshape(n, -1).sum(1)
                 for i in range(stride)]
        overlaps = torch.stack(parts, -1).sum(-1) / stride
    elif loss_type == 'bbox':
        overlaps = l_min.sum(dim=1) / l_max.sum(dim=1)
    else:  # keypoint: per (neg,pos) pair ratio, weighted by visibility, centre pair always counted
        n = pred.shape[0]
        mx = l_max.reshape(n, -1, 2).clamp(min=eps)
        mn = l_min.reshape(n, -1, 2)
        ratio = mn.sum(-1) / mx.sum(-1)                                   # (N, 2*(nv+1))
        vis = (vs > 0).to(ratio.dtype)
        vis = torch.stack((vis, vis), 2).reshape(n, -1)                   # (N, 2*nv)
        ratio = torch.cat([ratio[:, :-2] * vis, ratio[:, -2:]], dim=1)
        overlaps = ratio.sum(-1) / ratio.shape[1]

    if loss_type == 'keypoint':
        return 1 - overlaps

    box = get_bbox_from_extreme(pred, anchor_pts) if loss_type == 'bbox' else get_bbox_from_polygon(pred, anchor_pts)
    enclose_wh = (torch.max(box[:, 2:], bbox_gt[:, 2:]) - torch.min(box[:, :2], bbox_gt[:, :2])).clamp(min=0)
    c2 = enclose_wh[:, 0] ** 2 + enclose_wh[:, 1] ** 2 + eps
    w1, h1 = box[:, 2] - box[:, 0], box[:, 3] - box[:, 1] + eps
    w2, h2 = bbox_gt[:, 2] - bbox_gt[:, 0], bbox_gt[:, 3] - bbox_gt[:, 1] + eps
    rho2 = ((bbox_gt[:, 0] + bbox_gt[:, 2]) - (box[:, 0] + box[:, 2])) ** 2 / 4 + \
           ((bbox_gt[:, 1] + bbox_gt[:, 3]) - (box[:, 1] + box[:, 3])) ** 2 / 4
    v = (4 / math.pi ** 2) * torch.pow(torch.atan(w2 / h2) - torch.atan(w1 / h1), 2)
    return 1 - (overlaps - (rho2 / c2 + v ** 2 / (1 - overlaps + v)))


@LOSSES.register_module()
class CrossIOULoss(nn.Module):

    def __init__(self, eps=1e-6, reduction='mean', loss_weight=1.0, loss_type='bbox', alpha=0.2, stride=9):
        super().__init__()
        self.eps, self.reduction, self.loss_weight = eps, reduction, loss_weight
        self.loss_type, self.alpha, self.stride = loss_type, alpha, stride

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if weight is not None and weight.dim() > 1:
            assert weight.shape == pred.shape
            weight = weight.mean(-1)
        from ...ops import cross_iou as fused
        if fused.enabled() and fused.usable(pred, target, self.loss_type) and kwargs.get('pos_inds') is not None:
            from .utils import weight_reduce_loss
            rows = fused.cross_iou_bbox_rows(pred, target, kwargs['pos_inds'], kwargs['anchor_pts'], kwargs['bbox_gt'],
                                             weight, self.alpha, self.eps)      # already weighted
            return self.loss_weight * weight_reduce_loss(rows, None, reduction, avg_factor)
        if (fused.enabled() and kwargs.get('pos_inds') is not None
                and fused.rows_usable(pred, target, self.loss_type, self.stride, kwargs.get('pos_inds'), kwargs.get('anchor_pts'),
                                      kwargs.get('bbox_gt'), kwargs.get('vs'), weight)
                and (self.loss_type == 'keypoint' or (kwargs.get('anchor_pts') is not None and kwargs.get('bbox_gt') is not None))
                and (self.loss_type == 'polygon' or kwargs.get('vs') is not None)):
            from .utils import weight_reduce_loss
            rows = fused.cross_iou_rows(pred, target, kwargs['pos_inds'], self.loss_type, kwargs.get('anchor_pts'),
                                        kwargs.get('bbox_gt'), kwargs.get('vs'), weight, self.alpha, self.eps, self.stride)
            return self.loss_weight * weight_reduce_loss(rows, None, reduction, avg_factor)
        return self.loss_weight * cross_iou_loss(pred, target, weight, loss_type=self.loss_type, eps=self.eps,
                                                 reduction=reduction, avg_factor=avg_factor, alpha=self.alpha,
                                                 stride=self.stride, **kwargs)
