"""FocalLoss on the native sigmoid-focal-loss op (mmdet/models/losses/focal_loss.py:74-186).

For the common call (per-point weight, reduction 'mean' with an avg_factor) the whole
`elementwise loss * weight -> sum -> / avg_factor` chain is ONE reduction kernel
(lsn_sigmoid_focal_loss_sum) forward and ONE elementwise kernel backward."""
import torch.nn as nn

from ...ops import sigmoid_focal_loss as _sigmoid_focal_loss
from ...ops import sigmoid_focal_loss_sum
from ...ops.focal import sigmoid_focal_loss_level_sums
from ..builder import LOSSES
from .utils import weight_reduce_loss


def sigmoid_focal_loss(pred, target, weight=None, gamma=2.0, alpha=0.25, reduction='mean', avg_factor=None):
    per_row = weight is None or (weight.dim() == 1 and weight.size(0) == pred.size(0))
    if reduction == 'mean' and avg_factor is not None and per_row:
        return sigmoid_focal_loss_sum(pred, target, weight, gamma, alpha) / avg_factor
    loss = _sigmoid_focal_loss(pred, target, gamma, alpha)
    if weight is not None:
        if weight.shape != loss.shape:
            if weight.size(0) == loss.size(0):
                weight = weight.view(-1, 1)          # one weight per prior
            else:
                assert weight.numel() == loss.numel()  # flattened per-prior-per-class weights
                weight = weight.view(loss.size(0), -1)
        assert weight.ndim == loss.ndim
    return weight_reduce_loss(loss, weight, reduction, avg_factor)


@LOSSES.register_module()
class FocalLoss(nn.Module):

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert use_sigmoid is True, 'Only sigmoid focal loss supported now.'
        self.use_sigmoid, self.gamma, self.alpha = use_sigmoid, gamma, alpha
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        return self.loss_weight * sigmoid_focal_loss(pred, target, weight, gamma=self.gamma, alpha=self.alpha,
                                                     reduction=reduction, avg_factor=avg_factor)

    def forward_levels(self, pred, target, weight, B, num_level, avg_factor):
        """(L,): what `forward(pred_l, target_l, weight_l, avg_factor)` returns for the rows of every level l of LSHead's
        concatenated tensor (B images x N_all rows, levels back to back per image) -- one launch for all levels."""
        assert self.reduction == 'mean' and avg_factor is not None
        sums = sigmoid_focal_loss_level_sums(pred, target, weight, B, num_level, self.gamma, self.alpha)
        return self.loss_weight * (sums / avg_factor)
