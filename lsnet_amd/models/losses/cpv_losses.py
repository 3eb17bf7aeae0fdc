"""Losses of the corner-point-verification variant (LSCPVHead): corner heat-maps, corner offsets, box-level semantics.

* `GaussianFocalLoss` -- mmdet/models/losses/gaussian_focal_loss.py:8-89 (CornerNet's penalty-reduced focal loss on
  PROBABILITIES against a Gaussian-splatted target);
* `SEPFocalLoss`      -- mmdet/models/losses/focal_loss.py:45-71,189-221: sigmoid focal loss whose positive part is
  weighted per element and normalised by the sum of those weights, the negative part by `avg_factor`;
* `SmoothL1Loss`      -- mmdet/models/losses/smooth_l1_loss.py:8-90.

`SEPFocalLoss` is written with masks instead of the reference's boolean gathers: no data-dependent shapes, so no
device->host synchronisation inside the training step (same sums, same gradients)."""
import torch
import torch.nn as nn

from ..builder import LOSSES
from .utils import weighted_loss


@weighted_loss
def gaussian_focal_loss(pred, gaussian_target, alpha=2.0, gamma=4.0):
    eps = 1e-12
    pos_weights = gaussian_target.eq(1)
    neg_weights = (1 - gaussian_target).pow(gamma)
    pos_loss = -(pred + eps).log() * (1 - pred).pow(alpha) * pos_weights
    neg_loss = -(1 - pred + eps).log() * pred.pow(alpha) * neg_weights
    return pos_loss + neg_loss


@LOSSES.register_module()
class GaussianFocalLoss(nn.Module):

    def __init__(self, alpha=2.0, gamma=4.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.alpha, self.gamma, self.reduction, self.loss_weight = alpha, gamma, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        return self.loss_weight * gaussian_focal_loss(pred, target, weight, alpha=self.alpha, gamma=self.gamma,
                                                      reduction=reduction, avg_factor=avg_factor)


def separate_sigmoid_focal_loss(pred, target, weight=None, gamma=2.0, alpha=0.25, reduction='mean', avg_factor=None):
    """sum_pos(-log p (1-p)^gamma w alpha) / sum_pos(w)  +  sum_neg(-log(1-p) p^gamma (1-alpha)) / avg_factor;
    positives are target == 1, negatives target < 1; without positives only the second term."""
    p = pred.sigmoid()
    target = target.type_as(pred)
    pos, neg = target.eq(1), target.lt(1)
    one = torch.ones_like(p)
    pp = torch.where(pos, p, one)                                 # log(1) = 0 outside the mask, gradient 0 too
    pn = torch.where(neg, p, torch.zeros_like(p))
    w = torch.where(pos, weight, torch.zeros_like(p))
    pos_loss = (-torch.log(pp) * torch.pow(1 - pp, gamma) * w * alpha).sum()
    neg_loss = (-torch.log(1 - pn) * torch.pow(pn, gamma) * (1 - alpha)).sum()
    wsum = w.sum()
    pos_term = torch.where(pos.any(), pos_loss / torch.where(pos.any(), wsum, torch.ones_like(wsum)),
                           torch.zeros_like(pos_loss))
    return pos_term + neg_loss / avg_factor


@LOSSES.register_module()
class SEPFocalLoss(nn.Module):

    def __init__(self, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0, use_sigmoid=True):
        super().__init__()
        self.gamma, self.alpha, self.reduction, self.loss_weight = gamma, alpha, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        return self.loss_weight * separate_sigmoid_focal_loss(pred, target, weight, gamma=self.gamma,
                                                              alpha=self.alpha, avg_factor=avg_factor)


@weighted_loss
def smooth_l1_loss(pred, target, beta=1.0):
    assert beta > 0
    assert pred.size() == target.size() and target.numel() > 0
    diff = torch.abs(pred - target)
    return torch.where(diff < beta, 0.5 * diff * diff / beta, diff - 0.5 * beta)


@LOSSES.register_module()
class SmoothL1Loss(nn.Module):

    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        return self.loss_weight * smooth_l1_loss(pred, target, weight, beta=self.beta, reduction=reduction,
                                                 avg_factor=avg_factor, **kwargs)
